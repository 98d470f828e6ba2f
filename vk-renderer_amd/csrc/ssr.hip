// ssr.hip — stochastic Hi-Z screen-space reflections: programs "pdf_preintegrate",
// "sssr_trace", "sssr_filter", "sssr_blur".
//
// Reference: src/advanced_ssr.cpp:95-114,147-214,308-438 and
// shaders/advanced_ssr/{preintegrate,trace,filter,blur}.comp, include/screen_trace.glsl.
// Roofline: HBM in the compulsory-bytes model (10.3 / 16 / 14 B per full-res pixel), but the
// march is a chain of <= 80 dependent texel fetches per ray and the blur up to 23x23 taps,
// so trace is latency-bound and blur ALU/LDS-bound.
#include <cstdlib>
#include "vkr_host.hpp"
#include "hiz_march.hpp"
#include "ssr_sampling.hpp"

namespace vkr {

// ---- pdf_preintegrate (preintegrate.comp:45-65,77-84) ------------------------------------
__global__ __launch_bounds__(256) void k_pdf_preintegrate(Tex out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= out.w || y >= out.h) return;
  const int STEP_COUNT = 2000;
  const float a = (2.0f * ((float)x + 0.5f)) / (float)out.fw - 1.0f;
  const float b = ((float)y + 0.5f) / (float)out.fh;
  const float p = b - a, q = b + a;
  const float dt = 2.0f / (float)STEP_COUNT;
  float sum = 0.0f;
  for (int i = 0; i < STEP_COUNT; i++) {
    const float t = -1.0f + dt * ((float)i + 0.5f);
    const float L = p * t + q;
    const float nom = (1.0f - t) * L;
    const float denom = (1.0f + t * t) - (0.5f * L) * L;
    sum += (L > 0.0f) ? nom / (denom * denom) : 0.0f;
  }
  *texel_ptr<float>(out, x, y) = (2.0f / (float)STEP_COUNT) * sum;
}

// ---- sssr_trace ------------------------------------------------------------------------------
struct TraceArgs {
  Pyramid depth;       // view mips 0.. = image mips 1.. (advanced_ssr.cpp:186)
  Tex normal;          // downsampled normals, RG16_UNORM
  Tex material;        // full-res RGBA8_SRGB, bilinear at half-res uv
  Tex pdf;
  Tex out_ray, out_occ;
  const float4* halton;
  Mat4 normal_mat;
  Proj pr;
  uint32_t frame_random;
  float max_roughness;
  float horizon_d2;    // host-computed: smallest float x with sqrtf(x) >= 0.3f (trace.comp:257)
  // quotients that are the same for every ray, divided once on the host (IEEE, as the kernel would): 1 / screen_size,
  // 0.005 / screen_size (screen_trace.glsl:10,20 at most_detailed_mip 0) and zfar / (zfar - znear) (gbuffer_encode.glsl:77)
  f2 screen_size_inv, uv_offset_abs;
  float f_over_fn;
  // windowed (multi-GPU, vkr_sssr_trace_windowed): `normal` holds only rows [nrm_row0, nrm_row1) of the frame when the march
  // ends; a ray whose hit-normal footprint leaves them is stored as a provisional hit and its test is left to
  // vkr_sssr_validate — pend_mask (R8, every pixel) says which, pend_data (two float4 per pixel) keeps R and the hit uv
  int nrm_row0, nrm_row1;
  Tex pend_mask, pend_data;
  // vkr_sssr_trace_split: the frame-wide queue of the rays the head launch parks after park_after compacted rounds
  struct { uint4* records; uint32_t* counters; uint32_t capacity; } q;
  int park_after;
  // vkr_sssr_trace_windowed_head: the rows of the pyramid this rank holds before the all-gather has arrived — the window image's
  // own levels (local.count of them; `depth` then only tells the frame's extents and level count, it is not read)
  Pyramid local;
};

#define TP_VEC 5  // uint4 per parked-ray record (18 dwords used): 80-byte stride, conflict-free for consecutive rays
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // one ds_read_b128
VKR_DEV uint4 lds_load4(const uint4* p) { const u32x4 v = *(const u32x4*)p; return make_uint4(v.x, v.y, v.z, v.w); }
// four scalar stores that the backend merges into one ds_write_b128 (a vector built in the source from struct members
// makes the optimiser keep the struct in scratch memory)
VKR_DEV void lds_store4(uint4* p, uint4 v) { *p = v; }
VKR_DEV int wave_rank(uint64_t mask) {  // number of set bits of `mask` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// record: vector k holds members k, k + 5, k + 10, k + 15 of (origin 0-2, direction 3-5, inv_direction 6-8, normal 9-11,
// view_vec 12-14, t 15, h 16, mip | i << 8 17) — never four neighbours of the struct in one vector (see lds_store4)
VKR_DEV void pool_store(uint4* p, const RayConst& c, const RayState& s) {
  lds_store4(p + 0, make_uint4(__float_as_uint(c.origin.x), __float_as_uint(c.direction.z), __float_as_uint(c.normal.y), __float_as_uint(s.t)));
  lds_store4(p + 1, make_uint4(__float_as_uint(c.origin.y), __float_as_uint(c.inv_direction.x), __float_as_uint(c.normal.z), __float_as_uint(s.h)));
  lds_store4(p + 2, make_uint4(__float_as_uint(c.origin.z), __float_as_uint(c.inv_direction.y), __float_as_uint(c.view_vec.x), (uint32_t)(s.mip & 0xFF) | ((uint32_t)s.i << 8)));
  lds_store4(p + 3, make_uint4(__float_as_uint(c.direction.x), __float_as_uint(c.inv_direction.z), __float_as_uint(c.view_vec.y), 0u));
  lds_store4(p + 4, make_uint4(__float_as_uint(c.direction.y), __float_as_uint(c.normal.x), __float_as_uint(c.view_vec.z), 0u));
}
VKR_DEV void pool_load(const uint4* p, RayConst& c, RayState& s) {
  const uint4 a = lds_load4(p), b = lds_load4(p + 1), d = lds_load4(p + 2), e = lds_load4(p + 3), f = lds_load4(p + 4);
  c.origin = mk3(__uint_as_float(a.x), __uint_as_float(b.x), __uint_as_float(d.x));
  c.direction = mk3(__uint_as_float(e.x), __uint_as_float(f.x), __uint_as_float(a.y));
  c.inv_direction = mk3(__uint_as_float(b.y), __uint_as_float(d.y), __uint_as_float(e.y));
  c.normal = mk3(__uint_as_float(f.y), __uint_as_float(a.z), __uint_as_float(b.z));
  c.view_vec = mk3(__uint_as_float(d.z), __uint_as_float(e.z), __uint_as_float(f.z));
  s.t = __uint_as_float(a.w); s.h = __uint_as_float(b.w);
  s.mip = (int)(int8_t)(d.w & 0xFFu); s.i = (int)(d.w >> 8);
}
VKR_DEV void pool_store_state(uint4* p, const RayState& s) {
  uint32_t* w = (uint32_t*)p;
  w[3] = __float_as_uint(s.t); w[7] = __float_as_uint(s.h); w[11] = (uint32_t)(s.mip & 0xFF) | ((uint32_t)s.i << 8);
}
VKR_DEV void pool_load_result(const uint4* p, RayState& s) {
  const uint32_t* w = (const uint32_t*)p;
  s.t = __uint_as_float(w[3]); s.h = __uint_as_float(w[7]);
}

// One thread per ray in the prologue / epilogue; the march in between runs in rounds of
// TRACE_ROUND steps with the block's unfinished rays compacted in LDS between rounds, because
// rays of one 8x8 tile finish anywhere between 16 and 80 steps (mean 30, per-wave maximum mean 59
// at 4K: half the lanes idle without compaction).  Every ray executes exactly the shader's step
// sequence; only which lane executes it changes.
#ifndef TRACE_WY
#define TRACE_WY 1          // rows of 8x8 tiles per block: the block covers 32 x (8 TRACE_WY) pixels with 4 TRACE_WY waves
#endif
#ifndef TRACE_ROUND
#define TRACE_ROUND 16      // steps per compacted round
#endif
#define TRACE_PIN_ROUND 16  // round 0: the pinned steps, run by every ray
#define TRACE_THREADS (256 * TRACE_WY)

// LDS of a trace block: the parked rays of the block and the two lists of the compacted rounds
struct TracePool {
  uint4 ray[TRACE_THREADS * TP_VEC];  // record of ray (= owner thread) r: RayConst, t, h, mip | i << 8; word 15: parked slot + 1
  uint16_t list[2][TRACE_THREADS];
  int count[2];
  uint32_t parked_n, queue_base;
};

// park this thread's unfinished ray (one list reservation per wave: rank within the ballot)
VKR_DEV void pool_park(TracePool& pool, bool running, int tid, int lane, const RayConst& rc, const RayState& st) {
  const uint64_t rm = __ballot(running);
  int base = 0;
  if (lane == 0 && rm) base = atomicAdd(&pool.count[0], __popcll(rm));
  base = __builtin_amdgcn_readfirstlane(base);
  if (running) {
    pool.list[0][base + wave_rank(rm)] = (uint16_t)tid;
    pool_store(pool.ray + tid * TP_VEC, rc, st);
  }
}

// Compacted rounds: thread k of the block advances the k-th unfinished ray by TRACE_ROUND steps, at most max_rounds times.
// Returns the list that holds the rays still unfinished (pool.count[that] of them; none when the march ran to its end).
// LOCAL: a ray that needs a texel of the frame that is not in memory stops where it is (word 15 of its record: parked).
template <bool LOCAL>
VKR_DEV int trace_rounds(const MarchEnv& env, TracePool& pool, int tid, int lane, int max_rounds) {
  int cur = 0;
  for (int round = 0;; cur ^= 1, round++) {
    __syncthreads();
    const int n = pool.count[cur];
    if (n == 0 || round == max_rounds) break;
    if (tid == 0) pool.count[cur ^ 1] = 0;
    __syncthreads();
    bool more = false;
    int ray = 0;
    if (tid < n) {
      ray = pool.list[cur][tid];
      RayConst q;
      RayState rs;
      pool_load(pool.ray + ray * TP_VEC, q, rs);
      if (LOCAL) {
        int r = MARCH_MORE;
#pragma unroll 1
        for (int k = 0; k < TRACE_ROUND && r == MARCH_MORE; k++) r = march_step_ex<true, 15, true>(env, q, rs, 80);
        more = r == MARCH_MORE;
        if (r == MARCH_PARK) ((uint32_t*)(pool.ray + ray * TP_VEC))[15] = 1u;
      } else {
        more = true;
#pragma unroll 1
        for (int k = 0; k < TRACE_ROUND && more; k++) more = march_step<true, 15>(env, q, rs, 80);
      }
      pool_store_state(pool.ray + ray * TP_VEC, rs);
    }
    const uint64_t mm = __ballot(more);
    if (mm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&pool.count[cur ^ 1], __popcll(mm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (more) pool.list[cur ^ 1][base + wave_rank(mm)] = (uint16_t)ray;
    }
  }
  return cur;
}

// ---- frame-wide queue of parked rays (vkr_sssr_trace_split: head launch -> resume launch) -----------------------------------
// A ray the head launch does not finish is written here with everything the rest of its march and its epilogue need —
// 80 bytes: {origin, t} {direction, h} {pixel normal, mip | i << 8} {R, roughness} {pixel depth, lx | ly << 16, -, -} —
// and the resume launch loads 256 consecutive records per block: rays of many tiles side by side, so its waves start full
// where a tile's own stragglers would leave one nearly empty wave of their block alive for two more rounds.  inv_direction and view_vec are
// recomputed from the record by the very operations that made them (safe_inverse, reconstruct_view_vec: bit-identical).
#define TQ_VEC 5
#define TQ_FINISHED 0x80000000u  // in the mip | i << 8 word: the march has ended, only the epilogue is owed (its hit depth lies on rows not held)
VKR_DEV void queue_store(uint4* r, const RayConst& c, const RayState& s, f3 R, float roughness, float pixel_depth, int lx, int ly, bool finished) {
  r[0] = make_uint4(__float_as_uint(c.origin.x), __float_as_uint(c.origin.y), __float_as_uint(c.origin.z), __float_as_uint(s.t));
  r[1] = make_uint4(__float_as_uint(c.direction.x), __float_as_uint(c.direction.y), __float_as_uint(c.direction.z), __float_as_uint(s.h));
  r[2] = make_uint4(__float_as_uint(c.normal.x), __float_as_uint(c.normal.y), __float_as_uint(c.normal.z),
                    (uint32_t)(s.mip & 0xFF) | ((uint32_t)s.i << 8) | (finished ? TQ_FINISHED : 0u));
  r[3] = make_uint4(__float_as_uint(R.x), __float_as_uint(R.y), __float_as_uint(R.z), __float_as_uint(roughness));
  r[4] = make_uint4(__float_as_uint(pixel_depth), (uint32_t)lx | ((uint32_t)ly << 16), 0u, 0u);
}

// the part of trace.comp after the march (:94-139): validity tests, the two stores
template <bool WINDOWED>
VKR_DEV void trace_epilogue(const TraceArgs& a, const Tex& depth0, const RayConst& rc, const RayState& st, f3 R, float roughness, float pixel_depth, int lx, int ly) {
  const Proj pr = a.pr;
  const f2 tex_size = mk2((float)a.out_ray.fw, (float)a.out_ray.fh);
  const f3 ray_start = rc.origin;
  const f3 out_ray = madd(rc.origin, st.t, rc.direction);
  const float h = st.h;
  const f3 pixel_normal = rc.normal, view_vec = rc.view_vec;
  bool valid_hit = true;  // i <= 80 always (trace.comp:265)

  // trace.comp:94-118
  {
    const f2 ray_step = mk2(fabsf(out_ray.x - ray_start.x) * tex_size.x, fabsf(out_ray.y - ray_start.y) * tex_size.y);
    if (vmax(ray_step.x, ray_step.y) < 2.0f) valid_hit = false;
  }
  if (!WINDOWED) {
    if (valid_hit) {
      const f3 hnw = decode_normal(sample<FmtRG16U>(a.normal, xy(out_ray)));
      const f3 hit_normal = xyz(mul(a.normal_mat, mk4(hnw.x, hnw.y, hnw.z, 0.0f)));
      if (dot(hit_normal, R) > 0.0f || dot(pixel_normal, R) < 0.0f) valid_hit = false;
    }
    if (valid_hit) {
      const float hit_depth = sample<FmtD24>(depth0, xy(out_ray));
      const float hit_z = linearize_depth2_unorm(hit_depth, pr.znear, pr.zfar);
      const float ray_z = linearize_depth2(out_ray.z, pr.znear, pr.zfar);
      if (ray_z > hit_z + 0.3f || ray_z < hit_z - 0.1f) valid_hit = false;
    }
  } else {
    // The same conjunction with the tests that need nothing remote first (the pyramid is whole-frame): a ray is only left
    // pending when every other test has passed and its hit-normal footprint has a row outside the rows of `normal` held here.
    if (valid_hit && dot(pixel_normal, R) < 0.0f) valid_hit = false;
    if (valid_hit) {
      const float hit_depth = sample<FmtD24>(depth0, xy(out_ray));
      const float hit_z = linearize_depth2_unorm(hit_depth, pr.znear, pr.zfar);
      const float ray_z = linearize_depth2(out_ray.z, pr.znear, pr.zfar);
      if (ray_z > hit_z + 0.3f || ray_z < hit_z - 0.1f) valid_hit = false;
    }
    bool pending = false;
    if (valid_hit) {
      // rows of the bilinear footprint of texture(normal, hit uv): sample<>() clamps y0 and y0 + 1 to the frame
      const int y0 = f2i(floorf(cfma(out_ray.y, (float)a.normal.fh, -0.5f)));
      const int r0 = iclamp(y0, 0, a.normal.fh - 1), r1 = iclamp(y0 + 1, 0, a.normal.fh - 1);
      pending = r0 < a.nrm_row0 || r1 >= a.nrm_row1;
      if (!pending) {
        const f3 hnw = decode_normal(sample<FmtRG16U>(a.normal, xy(out_ray)));
        const f3 hit_normal = xyz(mul(a.normal_mat, mk4(hnw.x, hnw.y, hnw.z, 0.0f)));
        if (dot(hit_normal, R) > 0.0f) valid_hit = false;
      } else {
        float4* pd = texel_ptr<float4>(a.pend_data, 2 * lx, ly);
        pd[0] = make_float4(R.x, R.y, R.z, 0.0f);
        pd[1] = make_float4(out_ray.x, out_ray.y, 0.0f, 0.0f);
      }
    }
    *texel_ptr<uint8_t>(a.pend_mask, lx, ly) = pending ? 1u : 0u;
  }
  {  // RGBA16_UNORM store (advanced_ssr.cpp:62)
    uint2 o;
    o.x = float_to_unorm16(out_ray.x) | (float_to_unorm16(out_ray.y) << 16);
    o.y = float_to_unorm16(out_ray.z) | (float_to_unorm16(valid_hit ? pixel_depth : 1.0f) << 16);
    *texel_ptr<uint2>(a.out_ray, lx, ly) = o;
  }
  // trace.comp:123-139: (occlusion, pdf) into gtao.raw.  h was reset to 0 inside the march,
  // so the `no_occlusion` (h == -1) case of the reference never fires.
  {
    // Exact arithmetic: the PDF lookup is ill-conditioned near the singular texels of the LUT, so
    // its coordinates must match the reference sequence bit for bit (a 1e-7 perturbation there
    // moves raw.y by far more than 1e-3).
    const f3 w0 = -normalize(view_vec);
    const f3 slice_normal = normalize(cross(w0, R));
    const f3 normal_projected = pixel_normal - dot(pixel_normal, slice_normal) * slice_normal;
    const f3 X = normalize(cross(slice_normal, w0));
    const float n = VKR_PI / 2.0f - acosf(dot(normalize(normal_projected), X));
    float hh = acosf(h);
    hh = vmin(n + vmin(hh - n, VKR_PI / 2.0f), hh);
    const float pdf = sampleGGXdirPDF(a.pdf, w0, pixel_normal, R, roughness);
    const float occlusion = arc_occlusion(hh, n, length(normal_projected));
    const float result = is_nan(occlusion) ? 0.0f : occlusion;
    uint2 o;
    o.x = float_to_half_bits(result) | (float_to_half_bits(pdf) << 16);
    o.y = 0u;
    *texel_ptr<uint2>(a.out_occ, lx, ly) = o;
  }
}

VKR_DEV MarchEnv trace_env(const TraceArgs& a, const uint4* s_mip) {
  MarchEnv env;
  env.mip_table = s_mip;
  env.mip_count = a.depth.count;
  env.screen_size = mk2((float)a.depth.mip[0].fw, (float)a.depth.mip[0].fh);
  env.screen_size_inv = a.screen_size_inv;
  env.uv_offset_abs = a.uv_offset_abs;  // most_detailed_mip = 0
  env.pr = a.pr;
  env.horizon_d2 = a.horizon_d2;
  return env;
}

// the level table of a block: levels of `depth` (whole-frame extents), bases of `local` where the march is LOCAL
template <bool LOCAL>
VKR_DEV void stage_mip_tables(const TraceArgs& a, int tid, uint4* s_mip, uint2* s_win) {
  const int l = tid & 15;
  const Tex& f = a.depth.mip[l < a.depth.count ? l : 0];  // (a load from the kernel arguments: every lane, with the others)
  uint4 d = mip_descriptor(f);
  uint2 w = make_uint2(0u, 0u);
  if (LOCAL) {
    const Tex& t = a.local.mip[l < a.local.count ? l : 0];
    const uint64_t base = (uint64_t)t.p;
    d = make_uint4((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)t.pitch, (uint32_t)f.w | ((uint32_t)f.h << 16));
    w = make_uint2((uint32_t)t.oy, (uint32_t)t.h);
  }
  if (tid < 16) { s_mip[tid] = d; if (LOCAL) s_win[tid] = w; }
}

// LOCAL epilogue guard: does the hit-depth sample of trace.comp:111-117 (texture(depth, hit uv) on level 0) touch a row of the
// frame that is not held?  Evaluated only for rays that pass the tests before it (same order as trace_epilogue<true>).
VKR_DEV bool hit_depth_rows_missing(const TraceArgs& a, const RayConst& rc, const RayState& st, f3 R) {
  const f2 tex_size = mk2((float)a.out_ray.fw, (float)a.out_ray.fh);
  const f3 out_ray = madd(rc.origin, st.t, rc.direction);
  const f2 ray_step = mk2(fabsf(out_ray.x - rc.origin.x) * tex_size.x, fabsf(out_ray.y - rc.origin.y) * tex_size.y);
  if (vmax(ray_step.x, ray_step.y) < 2.0f) return false;
  if (dot(rc.normal, R) < 0.0f) return false;
  const Tex& l0 = a.local.mip[0];
  const int y0 = f2i(floorf(cfma(out_ray.y, (float)l0.fh, -0.5f)));
  const int r0 = iclamp(y0, 0, l0.fh - 1), r1 = iclamp(y0 + 1, 0, l0.fh - 1);
  return r0 < l0.oy || r1 >= l0.oy + l0.h;
}

// PARK: a head launch (vkr_sssr_trace_split, vkr_sssr_trace_windowed_head) — rays still unfinished after a.park_after
// compacted rounds leave for the frame-wide queue and their pixels are written by k_sssr_trace_resume.
// LOCAL (multi-GPU head): the pyramid is `a.local`, the rows of the first levels this rank computed itself; a ray is parked
// at its first fetch of a texel of the frame that is not there, and a finished ray whose hit-depth sample needs such rows too.
template <bool WINDOWED, bool PARK, bool LOCAL>
__global__ __launch_bounds__(TRACE_THREADS) void k_sssr_trace(TraceArgs a) {
  const i2 blk = xcd_block<4, 8 / TRACE_WY>();  // chunks of 128 x 64 output pixels
  __shared__ uint4 s_mip[16];
  __shared__ uint2 s_win[LOCAL ? 16 : 1];
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  __shared__ TracePool pool;
  const int tid = threadIdx.x;
  // wave w owns the 8x8 tile (blk.x*4 + w % 4, blk.y*TRACE_WY + w / 4)
  const int wave = tid >> 6, lane = tid & 63;
  const int lx = (blk.x * 4 + (wave & 3)) * 8 + (lane & 7);
  const int ly = (blk.y * TRACE_WY + (wave >> 2)) * 8 + (lane >> 3);
  const bool active = lx < a.out_ray.w && ly < a.out_ray.h;
  const int gx = a.out_ray.ox + lx, gy = a.out_ray.oy + ly;
  const f2 tex_size = mk2((float)a.out_ray.fw, (float)a.out_ray.fh);
  const f2 screen_uv = mk2(pixel_centre_uv(gx, tex_size.x), pixel_centre_uv(gy, tex_size.y));
  const Proj pr = a.pr;
  const Tex& depth0 = LOCAL ? a.local.mip[0] : a.depth.mip[0];
  // the pixel's three samples (trace.comp:49-58) are in flight before the block waits for its tables
  // (every lane: the texel indices are clamped into the images, and a conditional load would have to be waited for at once)
  const BilinearTaps taps_material = bilinear_taps_u32(a.material, screen_uv);
  const BilinearTaps taps_depth = bilinear_taps_u32(depth0, screen_uv);
  const BilinearTaps taps_normal = bilinear_taps_u32(a.normal, screen_uv);
  srgb_lut_stage(s_lut, tid, TRACE_THREADS);
  stage_mip_tables<LOCAL>(a, tid, s_mip, s_win);
  if (tid < 2) pool.count[tid] = 0;
  if (tid == 2) pool.parked_n = 0u;
  __syncthreads();

  MarchEnv env = trace_env(a, s_mip);
  if (LOCAL) { env.win_table = s_win; env.local_levels = a.local.count; }

  RayConst rc;
  RayState st;
  f3 R = mk3(0, 0, 0);
  float roughness = 0.0f, pixel_depth = 1.0f;
  bool running = false, parked = false;
  if (active) {
    // trace.comp:49-58
    roughness = taps_srgb_channel(taps_material, 1, s_lut);
    const float mg = mixf(0.0f, a.max_roughness, roughness);
    roughness = mg * mg;
    pixel_depth = taps_resolve<FmtD24>(taps_depth);
    const f3 pixel_normal_world = decode_normal(taps_resolve<FmtRG16U>(taps_normal));
    rc.normal = normalize(xyz(mul(a.normal_mat, mk4(pixel_normal_world.x, pixel_normal_world.y, pixel_normal_world.z, 0.0f))));
    rc.view_vec = reconstruct_view_vec(screen_uv, pixel_depth, pr);

    // trace.comp:61-63,156-158: rand() -> Halton index; sin evaluated in double
    const float rdot = dot(screen_uv, mk2(12.9898f, 78.233f));
    const float rnd01 = fractf(sin_hash_arg(rdot) * 43758.5453f);
    const uint32_t base_index = f2u(rnd01 * (float)VKR_HALTON_SEQ_SIZE);
    const uint32_t index = (base_index + a.frame_random) & (VKR_HALTON_SEQ_SIZE - 1);
    const float4 hv = a.halton[index];

    // trace.comp:65-77
    f3 tangent = get_tangent(rc.normal);
    const f3 bitangent = normalize(cross(rc.normal, tangent));
    tangent = normalize(cross(bitangent, rc.normal));
    f3 view_dir = -normalize(rc.view_vec);
    view_dir = mk3(dot(view_dir, tangent), dot(view_dir, bitangent), dot(view_dir, rc.normal));
    const f3 brdf_norm = sampleGGXVNDF(view_dir, roughness, roughness, hv.x, hv.z, hv.w);
    const f3 N = (brdf_norm.x * tangent + brdf_norm.y * bitangent) + brdf_norm.z * rc.normal;
    R = reflect(rc.view_vec, N);

    // trace.comp:79-84
    f3 ray_start = project_view_vec(rc.view_vec + 0.001f * rc.normal, pr, a.f_over_fn);
    ray_start.z -= 0.0001f;
    f3 ray_dir = project_view_vec(rc.view_vec + R, pr, a.f_over_fn);
    ray_dir = ray_dir - ray_start;
    ray_dir = ray_dir * ((1.0f - ray_start.z) / ray_dir.z);

    rc.origin = ray_start;
    rc.direction = ray_dir;
    rc.inv_direction = safe_inverse(ray_dir);
    st.t = initial_advance(env, rc);
    st.h = 0.0f;  // trace.comp:239
    st.mip = 0;
    st.i = 0;
    // round 0: the first 15 steps never leave mip 0 (specialised step), step 16 (i = 15) is the first that may
    if (!LOCAL) {
      const auto fetch0 = [&](int x, int y, float* z) -> bool {
        *z = ((uint32_t)x < (uint32_t)depth0.w && (uint32_t)y < (uint32_t)depth0.h) ? d24_to_float(*(const uint32_t*)(depth0.p + toff(depth0, x, y, 4))) : 0.0f;
        return true;
      };
#pragma unroll 1
      for (int k = 0; k < 15; k++) march_step_pinned0<false>(env, rc, st, fetch0);
      running = march_step<true, 15>(env, rc, st, 80);
    } else {
      // level 0 of the frame is depth0.fw x depth0.fh; rows [depth0.oy, + depth0.h) of it are here
      const auto fetch0 = [&](int x, int y, float* z) -> bool {
        *z = 0.0f;
        if ((uint32_t)x >= (uint32_t)depth0.fw || (uint32_t)y >= (uint32_t)depth0.fh) return true;  // outside the frame: 0
        const uint32_t row = (uint32_t)(y - depth0.oy);
        if (row >= (uint32_t)depth0.h) return false;
        *z = d24_to_float(*(const uint32_t*)(depth0.p + toff(depth0, x, (int)row, 4)));
        return true;
      };
      bool ok = true;
#pragma unroll 1
      for (int k = 0; k < 15 && ok; k++) ok = march_step_pinned0<true>(env, rc, st, fetch0);
      int r = MARCH_PARK;
      if (ok) r = march_step_ex<true, 15, true>(env, rc, st, 80);
      running = r == MARCH_MORE;
      parked = r == MARCH_PARK;
    }
  }
  pool_park(pool, running, tid, lane, rc, st);
  const int left = trace_rounds<LOCAL>(env, pool, tid, lane, PARK ? a.park_after : 1 << 30);
  bool finished_parked = false;  // LOCAL: the march has ended but its epilogue needs depth rows that are not here
  if (PARK) {
    const int n_left = pool.count[left];  // (uniform: nothing writes the counts after the rounds' last barrier)
    if (n_left) {
      if (tid < n_left) ((uint32_t*)(pool.ray + pool.list[left][tid] * TP_VEC))[15] = 1u;
      __syncthreads();
    }
    if (running) {  // this thread's ray was advanced by other lanes
      const uint32_t* w = (const uint32_t*)(pool.ray + tid * TP_VEC);
      pool_load_result(pool.ray + tid * TP_VEC, st);
      st.mip = (int)(int8_t)(w[11] & 0xFFu); st.i = (int)(w[11] >> 8);
      parked = w[15] != 0u;
    }
    if (LOCAL && active && !parked) { finished_parked = hit_depth_rows_missing(a, rc, st, R); parked = finished_parked; }
    // slots of the queue: one reservation per wave in LDS, one per block in the frame-wide counter
    const uint64_t pm = __ballot(parked);
    uint32_t wbase = 0u;
    if (lane == 0 && pm) wbase = atomicAdd(&pool.parked_n, (uint32_t)__popcll(pm));
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    __syncthreads();
    if (tid == 0 && pool.parked_n) pool.queue_base = atomicAdd(a.q.counters, pool.parked_n);
    __syncthreads();
    if (parked) {
      const uint32_t slot = pool.queue_base + wbase + (uint32_t)wave_rank(pm);
      if (slot < a.q.capacity) queue_store(a.q.records + (uint64_t)slot * TQ_VEC, rc, st, R, roughness, pixel_depth, lx, ly, finished_parked);
      return;
    }
  } else if (running) {
    pool_load_result(pool.ray + tid * TP_VEC, st);
  }
  if (!active) return;
  // (LOCAL: level 0 as this rank holds it — hit_depth_rows_missing() has sent every ray that needs other rows to the queue)
  trace_epilogue<WINDOWED>(a, depth0, rc, st, R, roughness, pixel_depth, lx, ly);
}

// The resume launch: every block takes 256 consecutive records of the queue (the rays of many tiles), marches them to their
// end and runs the epilogue for their pixels.  The record stays in the queue while its ray
// marches, so a thread holds nothing across the rounds but its record's index.
// COMPACT: the block's rays march in the compacted rounds of the head launch (the multi-GPU resume: most rays of the strip, long
// marches — throughput counts).  Without it every lane marches its own ray to the end: the single-GPU resume is a tenth of the
// rays with at most 32 steps left, 3 blocks per CU, bound by the latency of its dependent fetches — there the rounds' barriers
// and list bookkeeping cost what their better lane use saves (measured: 0.704 against 0.707 ms per frame, and 5 KB instead of 22 KB of LDS).
#define RESUME_BLOCK TRACE_THREADS
template <bool WINDOWED, bool COMPACT>
__global__ __launch_bounds__(RESUME_BLOCK) void k_sssr_trace_resume(TraceArgs a) {
  __shared__ uint4 s_mip[16];
  __shared__ uint4 pool_storage[COMPACT ? (sizeof(TracePool) + 15) / 16 : 1];  // (no pool without the rounds)
  TracePool& pool = *(TracePool*)pool_storage;
  __shared__ uint32_t s_total;
  const int tid = threadIdx.x, lane = tid & 63;
  stage_mip_tables<false>(a, tid, s_mip, nullptr);
  if (tid == 0) {
    const uint32_t n = *a.q.counters;  // written by the head launch's atomics; the launch boundary makes it visible
    s_total = n < a.q.capacity ? n : a.q.capacity;
  }
  __syncthreads();
  const uint32_t total = s_total;
  const MarchEnv env = trace_env(a, s_mip);
  const Proj pr = a.pr;
  const f2 tex_size = mk2((float)a.out_ray.fw, (float)a.out_ray.fh);
  for (uint32_t first = blockIdx.x * RESUME_BLOCK; first < total; first += gridDim.x * RESUME_BLOCK) {
    if (COMPACT) {
      if (tid < 2) pool.count[tid] = 0;
      __syncthreads();
    }
    const uint32_t idx = first + (uint32_t)tid;
    const bool have = idx < total;
    const uint4* rec = a.q.records + (uint64_t)idx * TQ_VEC;
    RayConst rc;
    RayState st;
    int lx = 0, ly = 0;
    float pixel_depth = 1.0f;
    bool finished = false;
    auto load_ray = [&]() {
      const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r4 = rec[4];
      rc.origin = mk3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
      rc.direction = mk3(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z));
      rc.inv_direction = safe_inverse(rc.direction);
      rc.normal = mk3(__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z));
      pixel_depth = __uint_as_float(r4.x);
      lx = (int)(r4.y & 0xFFFFu); ly = (int)(r4.y >> 16);
      const f2 screen_uv = mk2(pixel_centre_uv(a.out_ray.ox + lx, tex_size.x), pixel_centre_uv(a.out_ray.oy + ly, tex_size.y));
      rc.view_vec = reconstruct_view_vec(screen_uv, pixel_depth, pr);
      st.t = __uint_as_float(r0.w); st.h = __uint_as_float(r1.w);
      st.mip = (int)(int8_t)(r2.w & 0xFFu); st.i = (int)((r2.w & ~TQ_FINISHED) >> 8);
      finished = (r2.w & TQ_FINISHED) != 0u;
    };
    if (have) load_ray();
    const bool marching = have && !finished;
    if (COMPACT) {
      pool_park(pool, marching, tid, lane, rc, st);
      trace_rounds<false>(env, pool, tid, lane, 1 << 30);
    } else if (marching) {
      bool more = true;
#pragma unroll 1
      while (more) more = march_step<true, 15>(env, rc, st, 80);
    }
    if (have) {
      if (COMPACT) {
        load_ray();  // (again: cheaper than five more live vectors across the rounds)
        if (marching) pool_load_result(pool.ray + tid * TP_VEC, st);
      }
      const uint4 r3 = rec[3];
      trace_epilogue<WINDOWED>(a, a.depth.mip[0], rc, st, mk3(__uint_as_float(r3.x), __uint_as_float(r3.y), __uint_as_float(r3.z)), __uint_as_float(r3.w), pixel_depth, lx, ly);
    }
    if (COMPACT) __syncthreads();  // the pool is reused by the next chunk
  }
}

// ---- sssr_validate: the deferred hit-normal test of the windowed trace (trace.comp:103-109) -----------------------
// For every pixel the windowed trace left pending (its hit-normal footprint was not in memory yet): the same sample, the
// same dot product; a ray that fails becomes invalid (w = 1.0), exactly what the one-GPU trace would have stored.
__global__ __launch_bounds__(256) void k_sssr_validate(Tex rays, Tex pend_mask, Tex pend_data, Tex normal, Mat4 normal_mat, const uint32_t* skip_if_set) {
  if (skip_if_set && *skip_if_set != 0u) return;  // the hit normals are not all here (vkr_sssr_validate_unless)
  const int lx = blockIdx.x * 64 + threadIdx.x, ly = blockIdx.y * 4 + threadIdx.y;
  if (lx >= rays.w || ly >= rays.h) return;
  if (*texel_ptr<uint8_t>(pend_mask, lx, ly) == 0u) return;
  const float4* pd = texel_ptr<float4>(pend_data, 2 * lx, ly);
  const float4 Rv = pd[0], hv = pd[1];
  const f3 hnw = decode_normal(sample<FmtRG16U>(normal, mk2(hv.x, hv.y)));
  const f3 hit_normal = xyz(mul(normal_mat, mk4(hnw.x, hnw.y, hnw.z, 0.0f)));
  if (dot(hit_normal, mk3(Rv.x, Rv.y, Rv.z)) > 0.0f)
    *(uint16_t*)(const_cast<uint8_t*>(rays.p) + toff(rays, lx, ly, 8) + 6u) = (uint16_t)0xFFFFu;  // w = 1.0: not a hit (filter.comp:93-95)
}

// ---- sssr_filter (filter.comp:36-149, FULL_RES 0) -----------------------------------------------
struct FilterArgs {
  Tex rays, depth1, albedo, normal, material, out;
  Mat4 normal_mat;
  Proj pr;
  uint32_t render_flags;
  uint32_t skip_empty_tiles;  // 1: a tile without a single hit writes zeros and returns
};

// Everything process_pixel() derives from the *tap* pixel alone (its ray, depth, normal, hit colour:
// filter.comp:112-134 and the Fresnel power term / N.L / N.V of ray_weight :97-103) is computed once
// per pixel into an LDS tile with a one-pixel apron; the five cross taps of every centre then only
// combine those with the centre's F0 / roughness / depth.  Same operations in the same order as the
// shader, each evaluated once instead of five times.
#define FILT_BX 32
#define FILT_BY 8
#define FILT_TW (FILT_BX + 2)
#define FILT_TH (FILT_BY + 2)

__global__ __launch_bounds__(FILT_BX * FILT_BY) void k_sssr_filter(FilterArgs a) {
  const i2 blk = xcd_block<4, 8>();  // chunks of 128 x 64 output pixels
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  __shared__ float4 s_geo[FILT_TW * FILT_TH];  // {fresnel power term, NdotL, NdotV, depth}
  __shared__ float4 s_rad[FILT_TW * FILT_TH];  // radiance rgb
  const int tid = threadIdx.y * FILT_BX + threadIdx.x;

  const f2 tex_size = mk2((float)a.out.fw, (float)a.out.fh);
  const int bx0 = a.out.ox + blk.x * FILT_BX - 1, by0 = a.out.oy + blk.y * FILT_BY - 1;
  // A tile none of whose rays (apron included) found a hit has radiance 0 at every tap: each pixel's colour sums stay
  // exactly 0 and the stored texel is 0 whatever the weights are (52 % of the tiles of the benchmark frame).  A ray
  // texel outside the frame reads 0, i.e. w != 1: it counts as a hit, as in the shader.
  // (the sRGB table is staged behind the rays' loads of the check and shares their wait and barrier)
  if (a.skip_empty_tiles) {
    bool hit = false;
    for (int t = tid; t < FILT_TW * FILT_TH; t += FILT_BX * FILT_BY) hit = hit || fetch<FmtRGBA16U>(a.rays, bx0 + t % FILT_TW, by0 + t / FILT_TW).w != 1.0f;
    srgb_lut_stage(s_lut, tid, FILT_BX * FILT_BY);
    if (__syncthreads_or(hit) == 0) {
      const int lx = blk.x * FILT_BX + threadIdx.x, ly = blk.y * FILT_BY + threadIdx.y;
      if (lx < a.out.w && ly < a.out.h) *texel_ptr<uint32_t>(a.out, lx, ly) = 0u;
      return;
    }
  } else {
    srgb_lut_stage(s_lut, tid, FILT_BX * FILT_BY);
    __syncthreads();
  }
  for (int t = tid; t < FILT_TW * FILT_TH; t += FILT_BX * FILT_BY) {
    const int px = bx0 + t % FILT_TW, py = by0 + t / FILT_TW;
    const f4 trace_result = fetch<FmtRGBA16U>(a.rays, px, py);
    const f2 pixel_uv = mk2((float)px / tex_size.x, (float)py / tex_size.y);
    const float pixel_depth = fetch<FmtD24>(a.depth1, px, py);
    const f3 view_vec = reconstruct_view_vec(pixel_uv, pixel_depth, a.pr);
    const f3 pnw = decode_normal_fast(sample<FmtRG16U>(a.normal, pixel_uv));
    const f3 Nn = xyz(mul(a.normal_mat, mk4(pnw.x, pnw.y, pnw.z, 0.0f)));
    const f3 hit_vec = reconstruct_view_vec(mk2(trace_result.x, trace_result.y), trace_result.z, a.pr);
    const f3 radiance = (trace_result.w != 1.0f) ? sample_srgb_rgb(a.albedo, mk2(trace_result.x, trace_result.y), s_lut) : mk3(0, 0, 0);
    // weights only (never compared): hardware rsq and x^5 by multiplication
    const f3 V = -normalize_fast(view_vec);
    const f3 L = normalize_fast(hit_vec - view_vec);
    const f3 H = normalize_fast(V + L);
    const float om = vclamp(1.0f - vmax(dot(H, V), 0.0f), 0.0f, 1.0f);
    const float om2 = om * om;
    const float p5 = (om2 * om2) * om;  // fresnelSchlick's pow(., 5)
    s_geo[t] = make_float4(p5, vmax(dot(Nn, L), 0.0f), vmax(dot(Nn, V), 0.0f), pixel_depth);
    s_rad[t] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
  }
  __syncthreads();

  const int lx = blk.x * FILT_BX + threadIdx.x;
  const int ly = blk.y * FILT_BY + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  const f2 screen_uv = mk2((float)gx / tex_size.x, (float)gy / tex_size.y);  // no +0.5 (filter.comp:39)
  const float metallic = sample_srgb_channel(a.material, screen_uv, 2, s_lut);
  const float roughness = sample_srgb_channel(a.material, screen_uv, 1, s_lut);
  const f3 albedo = sample_srgb_rgb(a.albedo, screen_uv, s_lut);
  const f3 F0 = F0_approximation(albedo, metallic);
  const f3 one_minus_F0 = mk3(1.0f, 1.0f, 1.0f) - F0;
  const float alpha2 = roughness * roughness;
  f3 color_sum = mk3(0, 0, 0), weight_sum = mk3(0, 0, 0);
  const int tc = (threadIdx.y + 1) * FILT_TW + (threadIdx.x + 1);
  const float center_depth = s_geo[tc].w;
  const float k_bilateral = 1000.0f * fast_rcp(center_depth);
  const int taps = (a.render_flags & VKR_NORMALIZE_REFLECTIONS) ? 5 : 1;
  const int offs[5] = {0, -1, FILT_TW, 1, -FILT_TW};  // (0,0) (-1,0) (0,1) (1,0) (0,-1): filter.comp:62-68
#pragma unroll
  for (int k = 0; k < 5; k++) {
    if (k < taps) {
      const float4 geo = s_geo[tc + offs[k]];
      const float4 rad = s_rad[tc + offs[k]];
      // ray_weight (filter.comp:97-108), literal swapped argument order of brdfG1
      const f3 F = F0 + one_minus_F0 * geo.x;
      // brdfG2(NdotL, NdotV, alpha2) / brdfG1(NdotV, alpha2) with hardware rcp / sqrt (smooth weights)
      const float NdotL2 = geo.y * geo.y, NdotV2 = geo.z * geo.z;
      const float L1 = fast_sqrt(1.0f + (alpha2 * (1.0f - NdotL2)) * fast_rcp(NdotL2));
      const float L2 = fast_sqrt(1.0f + (alpha2 * (1.0f - NdotV2)) * fast_rcp(NdotV2));
      const float G2 = 2.0f * fast_rcp(L1 + L2);
      const float a4 = alpha2 * alpha2;  // brdfG1's "NdotV" argument is alpha2 here (filter.comp:106)
      const float G1 = 2.0f * fast_rcp(1.0f + fast_sqrt(1.0f + geo.z * ((1.0f - a4) * fast_rcp(a4))));
      f3 weight = F * (G2 * fast_rcp(G1));
      float bilateral_weight = 1.0f;
      if (a.render_flags & VKR_BILATERAL_FILTER)  // continuous weight: one reciprocal of the centre depth for the five taps
        bilateral_weight = vmax(__builtin_fmaf(-fabsf(center_depth - geo.w), k_bilateral, 1.0f), 0.0f);
      weight = weight * bilateral_weight;
      color_sum = color_sum + weight * mk3(rad.x, rad.y, rad.z);
      weight_sum = weight_sum + weight;
    }
  }
  if (vmax(weight_sum.x, vmax(weight_sum.y, weight_sum.z)) < 0.001f) weight_sum = mk3(1, 1, 1);
  color_sum = mk3(color_sum.x * fast_rcp(weight_sum.x), color_sum.y * fast_rcp(weight_sum.y), color_sum.z * fast_rcp(weight_sum.z));
  *texel_ptr<uint32_t>(a.out, lx, ly) =
      float_to_unorm8(color_sum.x) | (float_to_unorm8(color_sum.y) << 8) | (float_to_unorm8(color_sum.z) << 16);
}

// ---- sssr_blur (blur.comp:31-115) ------------------------------------------------------------------
struct BlurArgs {
  Tex depth1, normal, refl, material, history, velocity, hist_depth1, out;
  Mat4 inverse_camera, prev_inverse_camera;
  Proj pr;
  float max_roughness;
  uint32_t accumulate, disable_blur;
  uint32_t skip_empty_tiles;  // 1: a tile whose staged reflections are all black skips its taps (the sums are exactly 0)
  uint32_t uniform_sigma_path;  // 1: waves whose pixels share one sigma take blur_uniform_sigma
  uint32_t rows_path;           // 1: every other wave takes blur_rows (0: the per-lane loops, VKR_SWITCH_BLUR_LANE_LOOPS)
};

// Tile geometry of the blur: a block resolves BLUR_BX x BLUR_BY pixels and stages the pixels within
// the largest possible radius (r <= 11: sigma <= 4, blur.comp:45,53) in LDS.  Every staged pixel
// is decoded once per block — depth (D24 -> float), normal (bilinear of 4 full-res texels,
// octahedral decode, normalise; blur.comp:63) and reflection colour — instead of once per tap:
// the reference shader repeats that work up to 529 times per pixel.
//
// The kernel is VALU-bound (rocprof: 98 % VALU-active), so the tap loop is built around packed-f32
// instructions: a thread resolves the two vertically adjacent pixels A = (x, 2y), B = (x, 2y+1) in
// the two lanes of v_pk_* ops; every staged tap is read once (ds_read_b128 + ds_read_b32) and
// broadcast to both lanes.  Summation order per pixel is the shader's (i outer, j inner).
// The tap weights are smooth functions, so they may differ from the shader's expression by
// rounding noise (~1e-6 against a 1e-3 / one-UNORM8-step tolerance): exp(-(i^2+j^2)/e) is the
// product of two running Gaussians advanced by recurrence (E(k+1) = E(k)*rho(k), rho(k+1) =
// rho(k)*exp(-2/e)), the bilateral term uses a reciprocal-multiply, colour is accumulated in
// UNORM8 code units and scaled by 1/255 once.
#define BLUR_BX 32
#ifndef BLUR_BY
#define BLUR_BY 32
#endif
#define BLUR_R 11
#define BLUR_TW (BLUR_BX + 2 * BLUR_R)
#define BLUR_TH (BLUR_BY + 2 * BLUR_R)
#define BLUR_THREADS (BLUR_BX * BLUR_BY / 2)
#ifndef BLUR_AHEAD
#define BLUR_AHEAD 3
#endif
#ifndef BLUR_WAVES
#define BLUR_WAVES 4  // waves per SIMD the register budget is sized for
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
VKR_DEV v2f splat2(float a) { return (v2f){a, a}; }
VKR_DEV v2f exp2_2(v2f x) { return (v2f){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

struct BlurCentre {  // per output pixel
  f3 normal;
  float depth, k_bilateral, g, neg_inv_e_log2;
  int r, tc;
};

// Development instrumentation, compiled only with -DVKR_BLUR_STAMPS (tools/blur_timeline.py builds such a library next to the
// product one): wave 0 of every block records the 100 MHz wall clock at its phase boundaries, plus where it ran.
#ifdef VKR_BLUR_STAMPS
__device__ unsigned long long g_blur_stamps[65536 * 8];
#define VKR_STAMP(slot) do { if (threadIdx.x == 0 && threadIdx.y == 0) g_blur_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#define VKR_STAMP_VALUE(slot, v) do { if (threadIdx.x == 0 && threadIdx.y == 0) g_blur_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = (unsigned long long)(v); } while (0)
#else
#define VKR_STAMP(slot) do {} while (0)
#define VKR_STAMP_VALUE(slot, v) do {} while (0)
#endif

// A staged pixel is ONE 16-byte record, so a tap is one ds_read_b128: {normal.x, normal.y, normal.z, depth} as floats, with the low
// mantissa byte of the three normal components replaced by the R, G, B codes of the reflection texel.  v_cvt_f32_ubyte0 reads a
// code straight out of the word; the normal keeps 15 mantissa bits (relative error < 2^-15), and it only enters the smooth normal
// weight.  The depth word is exact: the bilateral weight is a cliff.
VKR_DEV uint4 blur_pack(f3 n, float depth, uint32_t rgba) {
  return make_uint4((__float_as_uint(n.x) & ~0xFFu) | (rgba & 0xFFu), (__float_as_uint(n.y) & ~0xFFu) | ((rgba >> 8) & 0xFFu),
                    (__float_as_uint(n.z) & ~0xFFu) | ((rgba >> 16) & 0xFFu), __float_as_uint(depth));
}

// accumulators: xyz = sum w * colour (UNORM8 code units), w = sum w
VKR_DEV void blur_tap(const uint4* s_px, const BlurCentre& c, int t, float wg, f4& acc) {
  const uint4 p = lds_load4(&s_px[t]);
  const float nx = __uint_as_float(p.x), ny = __uint_as_float(p.y), nz = __uint_as_float(p.z), d = __uint_as_float(p.w);
  const float bil = vmax(__builtin_fmaf(-fabsf(c.depth - d), c.k_bilateral, 1.0f), 0.0f);
  const float nw = vmax(__builtin_fmaf(c.normal.z, nz, __builtin_fmaf(c.normal.y, ny, c.normal.x * nx)), 0.0f);
  const float w = (wg * bil) * nw;
  acc.x = __builtin_fmaf(w, (float)(p.x & 0xFFu), acc.x);
  acc.y = __builtin_fmaf(w, (float)(p.y & 0xFFu), acc.y);
  acc.z = __builtin_fmaf(w, (float)(p.z & 0xFFu), acc.z);
  acc.w += w;
}

// one pixel on its own (rows of the pair with different radii, or a missing partner row)
VKR_DEV f4 blur_single(const uint4* s_px, const BlurCentre& c) {
  f4 acc = mk4(0, 0, 0, 0);
  const float kappa = __builtin_amdgcn_exp2f(2.0f * c.neg_inv_e_log2);  // exp(-2/e)
  const int r = c.r;
  // E(-r) and rho(-r) = exp(-(2*(-r)+1)/e)
  const float e_start = __builtin_amdgcn_exp2f((float)(r * r) * c.neg_inv_e_log2);
  const float rho_start = __builtin_amdgcn_exp2f((float)(1 - 2 * r) * c.neg_inv_e_log2);
  float ei = e_start, rho_i = rho_start;
#pragma unroll 1
  for (int i = -r; i <= r; i++) {
    const float gi = c.g * ei;
    float ej = e_start, rho_j = rho_start;
#pragma unroll 1
    for (int j = -r; j <= r; j++) {
      blur_tap(s_px, c, c.tc + i + j * BLUR_TW, gi * ej, acc);
      ej *= rho_j; rho_j *= kappa;
    }
    ei *= rho_i; rho_i *= kappa;
  }
  return acc;
}

// Wave-uniform sigma (a material's roughness is constant over most of a surface, so the 32 x 4 pixels of a wave usually share
// it): the Gaussian factors E(k) = exp(-k^2 / e) are then the same numbers in every lane.  They are evaluated once, held as the
// R + 1 register pairs P[k] = {E(k), E(k+1)} (E(R+1) := 0), and a staged row s — row s of pixel A, row s - 1 of pixel B — takes
// its pair of row factors {E(|s|), E(|s-1|)} from one of them, swapped for s >= 1 by the operand selectors of the packed
// multiply.  Against the general loop below: no running Gaussian products (two packed multiplies per tap), the rows of a column
// fully unrolled (LDS offsets are immediates, no loop counter), the column factor E(i) applied once per column to the column's
// sums, the unpaired end rows served by the zero in P[R].  Per pixel the taps and their weights are the shader's; the sums are
// formed column by column (rounding noise against a 1e-3 / one-code tolerance).
template <bool SWAP> VKR_DEV v2f pk_mul_rows(v2f a, v2f p) {
  if (!SWAP) return a * p;
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(p));
  return r;
}

template <int R> VKR_DEV void blur_uniform_sigma(const uint4* s_px, const BlurCentre* c, f4& accA, f4& accB) {
  const float nie = c[0].neg_inv_e_log2, g = c[0].g;  // the same in every lane and for both pixels
  v2f P[R + 1];
#pragma unroll
  for (int k = 0; k <= R; k++)
    P[k] = (v2f){__builtin_amdgcn_exp2f((float)(k * k) * nie), k < R ? __builtin_amdgcn_exp2f((float)((k + 1) * (k + 1)) * nie) : 0.0f};
  const v2f cnx = {c[0].normal.x, c[1].normal.x}, cny = {c[0].normal.y, c[1].normal.y}, cnz = {c[0].normal.z, c[1].normal.z};
  const v2f kb = {c[0].k_bilateral, c[1].k_bilateral};
  const v2f kcd = kb * (v2f){c[0].depth, c[1].depth};
  const float kappa = __builtin_amdgcn_exp2f(2.0f * nie);
  float ei = P[R].x, rho_i = __builtin_amdgcn_exp2f((float)(1 - 2 * R) * nie);  // E(-R), rho(-R)
  v2f acc_r = {0.0f, 0.0f}, acc_g = {0.0f, 0.0f}, acc_b = {0.0f, 0.0f}, acc_w = {0.0f, 0.0f};
  const uint4* col = s_px + (c[0].tc - R - R * BLUR_TW);  // row -R of pixel A, column -R
#pragma unroll 1
  for (int i = -R; i <= R; i++, col++) {
    v2f col_r = {0.0f, 0.0f}, col_g = {0.0f, 0.0f}, col_b = {0.0f, 0.0f}, col_w = {0.0f, 0.0f};
    uint4 ahead[BLUR_AHEAD];  // the reads run BLUR_AHEAD rows ahead of the arithmetic
#pragma unroll
    for (int k = 0; k < BLUR_AHEAD; k++) ahead[k] = lds_load4(col + k * BLUR_TW);
#pragma unroll
    for (int s = -R; s <= R + 1; s++) {
      const uint4 p = ahead[(s + R) % BLUR_AHEAD];
      if (s + R + BLUR_AHEAD <= 2 * R + 1) ahead[(s + R) % BLUR_AHEAD] = lds_load4(col + (s + R + BLUR_AHEAD) * BLUR_TW);
      const v2f nzw = {__uint_as_float(p.z), __uint_as_float(p.w)};
      const v2f nxy = __builtin_elementwise_fma(cny, splat2(__uint_as_float(p.y)), cnx * splat2(__uint_as_float(p.x)));
      v2f nw, kdz;
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(nw) : "v"(cnz), "v"(nzw), "v"(nxy));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
          : "=v"(kdz) : "v"(kb), "v"(nzw), "v"(kcd));
      const v2f tw = s >= 1 ? pk_mul_rows<true>(nw, P[s >= 1 ? s - 1 : 0]) : pk_mul_rows<false>(nw, P[s >= 1 ? 0 : -s]);
      v2f w;
      w.x = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.x), tw.x, tw.x), 0.0f), 1.0f);
      w.y = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.y), tw.y, tw.y), 0.0f), 1.0f);
      col_r = __builtin_elementwise_fma(w, splat2((float)(p.x & 0xFFu)), col_r);
      col_g = __builtin_elementwise_fma(w, splat2((float)(p.y & 0xFFu)), col_g);
      col_b = __builtin_elementwise_fma(w, splat2((float)(p.z & 0xFFu)), col_b);
      col_w += w;
    }
    const v2f e2 = splat2(ei);
    acc_r = __builtin_elementwise_fma(e2, col_r, acc_r);
    acc_g = __builtin_elementwise_fma(e2, col_g, acc_g);
    acc_b = __builtin_elementwise_fma(e2, col_b, acc_b);
    acc_w = __builtin_elementwise_fma(e2, col_w, acc_w);
    ei *= rho_i; rho_i *= kappa;
  }
  accA = mk4(acc_r.x * g, acc_g.x * g, acc_b.x * g, acc_w.x * g);
  accB = mk4(acc_r.y * g, acc_g.y * g, acc_b.y * g, acc_w.y * g);
}

// Per-lane sigma (a material texture: roughness, hence sigma and radius, differ from pixel to pixel): the same packed tap as
// blur_uniform_sigma, with the loop nest TRANSPOSED so that per-pixel Gaussian factors still fit a small register table.  Pixels
// A and B of a thread are vertically adjacent, so in one staged ROW they share the column index i: the column factors of a row
// are the pairs C[k] = {E_A(k), E_B(k)}, k = |i| — R + 1 register pairs, evaluated once per thread from its two sigmas — and the
// 2R + 1 columns of a row are unrolled (consecutive LDS records, immediate offsets, reads running ahead); the row factor
// {E_A(|s|), E_B(|s - 1|)} is applied once per row to the row's sums.  R is the largest radius in the WAVE: a pixel with a smaller
// radius has zeros in its table entries beyond it (and in its row factors), so its extra taps add exactly 0 and no lane
// diverges.  Replaces the per-lane loops of rounds 1-3 (11 VALU instructions per pixel-tap on divergent trip counts, pairs of
// different radius resolved one pixel at a time) where a wave does not share one sigma: 7 per pixel-tap.  Per pixel the taps and
// weights are the shader's; the sums are formed row by row.
template <int R> VKR_DEV void blur_rows(const uint4* s_px, const BlurCentre* c, bool has_b, f4& accA, f4& accB) {
  const int rA = c[0].r, rB = has_b ? c[1].r : -1;
  const v2f nie = {c[0].neg_inv_e_log2, c[1].neg_inv_e_log2};
  v2f C[R + 1];
#pragma unroll
  for (int k = 0; k <= R; k++) {
    const v2f e = exp2_2((float)(k * k) * nie);
    C[k] = (v2f){k <= rA ? e.x : 0.0f, k <= rB ? e.y : 0.0f};
  }
  const v2f cnx = {c[0].normal.x, c[1].normal.x}, cny = {c[0].normal.y, c[1].normal.y}, cnz = {c[0].normal.z, c[1].normal.z};
  const v2f kb = {c[0].k_bilateral, c[1].k_bilateral};
  const v2f kcd = kb * (v2f){c[0].depth, c[1].depth};
  v2f acc_r = {0.0f, 0.0f}, acc_g = {0.0f, 0.0f}, acc_b = {0.0f, 0.0f}, acc_w = {0.0f, 0.0f};
  const uint4* row = s_px + (c[0].tc - R - R * BLUR_TW);  // row -R of pixel A, column -R
#pragma unroll 1
  for (int s = -R; s <= R + 1; s++, row += BLUR_TW) {
    // row s of A is row s - 1 of B
    const int ja = s < 0 ? -s : s, jb = s < 1 ? 1 - s : s - 1;
    v2f f = exp2_2((v2f){(float)(ja * ja), (float)(jb * jb)} * nie);
    f = (v2f){ja <= rA ? f.x : 0.0f, jb <= rB ? f.y : 0.0f};
    v2f row_r = {0.0f, 0.0f}, row_g = {0.0f, 0.0f}, row_b = {0.0f, 0.0f}, row_w = {0.0f, 0.0f};
    uint4 ahead[BLUR_AHEAD];
#pragma unroll
    for (int k = 0; k < BLUR_AHEAD; k++) ahead[k] = lds_load4(row + k);
#pragma unroll
    for (int i = -R; i <= R; i++) {
      const uint4 p = ahead[(i + R) % BLUR_AHEAD];
      if (i + R + BLUR_AHEAD <= 2 * R) ahead[(i + R) % BLUR_AHEAD] = lds_load4(row + (i + R + BLUR_AHEAD));
      const v2f nzw = {__uint_as_float(p.z), __uint_as_float(p.w)};
      const v2f nxy = __builtin_elementwise_fma(cny, splat2(__uint_as_float(p.y)), cnx * splat2(__uint_as_float(p.x)));
      v2f nw, kdz;
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(nw) : "v"(cnz), "v"(nzw), "v"(nxy));
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
          : "=v"(kdz) : "v"(kb), "v"(nzw), "v"(kcd));
      const v2f tw = nw * C[i < 0 ? -i : i];
      v2f w;
      w.x = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.x), tw.x, tw.x), 0.0f), 1.0f);
      w.y = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.y), tw.y, tw.y), 0.0f), 1.0f);
      row_r = __builtin_elementwise_fma(w, splat2((float)(p.x & 0xFFu)), row_r);
      row_g = __builtin_elementwise_fma(w, splat2((float)(p.y & 0xFFu)), row_g);
      row_b = __builtin_elementwise_fma(w, splat2((float)(p.z & 0xFFu)), row_b);
      row_w += w;
    }
    acc_r = __builtin_elementwise_fma(f, row_r, acc_r);
    acc_g = __builtin_elementwise_fma(f, row_g, acc_g);
    acc_b = __builtin_elementwise_fma(f, row_b, acc_b);
    acc_w = __builtin_elementwise_fma(f, row_w, acc_w);
  }
  const float gA = c[0].g, gB = c[1].g;
  accA = mk4(acc_r.x * gA, acc_g.x * gA, acc_b.x * gA, acc_w.x * gA);
  accB = mk4(acc_r.y * gB, acc_g.y * gB, acc_b.y * gB, acc_w.y * gB);
}

// one BLUR_BX x BLUR_BY tile of the output; s_px: the staged tile, one 16-byte record per pixel (blur_pack); s_lut: the sRGB
// decode table (storage; staged inside, visible after the first barrier)
VKR_DEV void blur_tile(const BlurArgs& a, const i2 blk, uint4* s_px, float* s_lut, const int tid) {
  const int bx0 = a.out.ox + blk.x * BLUR_BX - BLUR_R;  // frame coordinates of the tile origin
  const int by0 = a.out.oy + blk.y * BLUR_BY - BLUR_R;
  const f2 tex_size = mk2((float)a.out.fw, (float)a.out.fh);

  // Global loads are issued in batches and decoded afterwards, so that a wave pays one memory latency per
  // batch instead of one per sample: (1) the velocity of its two pixels, (2) the raw texels of the <= 6 tile
  // pixels it stages, (3) everything else its two pixels need (roughness, centre normal, the reprojection
  // test's depths, the history colour).  The tap loop and the epilogue then run from registers and LDS only.
  const int lx = blk.x * BLUR_BX + threadIdx.x;
  const int lyA = blk.y * BLUR_BY + 2 * threadIdx.y;
  const bool live = lx < a.out.w && lyA < a.out.h;  // (whole blocks past the window do not exist; rows / columns may)
  const bool has_b = lyA + 1 < a.out.h;
  f2 uv_c[2];
  BilinearTaps velocity_taps[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int gx = a.out.ox + min(lx, a.out.w - 1), gy = a.out.oy + min(lyA + k, a.out.h - 1);
    uv_c[k] = mk2(pixel_centre_uv(gx, tex_size.x), pixel_centre_uv(gy, tex_size.y));
    velocity_taps[k] = bilinear_taps_u32(a.velocity, uv_c[k]);
  }

  constexpr int STAGE_ITERS = (BLUR_TW * BLUR_TH + BLUR_THREADS - 1) / BLUR_THREADS;
  // The reflections of the tile first: when every staged texel is black (no valid hit within the apron: sky, rough or
  // unlit regions — 46 % of the tiles of the benchmark frame) each tap adds w * 0 to the colour sums, the pixel's result
  // is 0 / max(sum w, 0.001) = 0 whatever the weights are, and the block skips the normal / depth staging and the taps.
  uint32_t stage_refl[STAGE_ITERS];
  uint32_t lit = 0u;
#pragma unroll
  for (int k = 0; k < STAGE_ITERS; k++) {
    const int t = min(tid + k * BLUR_THREADS, BLUR_TW * BLUR_TH - 1);  // the last batch re-stages the last pixel
    const int tx = t % BLUR_TW, ty = t / BLUR_TW;
    const int px = bx0 + tx, py = by0 + ty;
    // texelFetch out of the frame -> 0 (the loads themselves are clamped into the window)
    const bool in_refl = px >= 0 && py >= 0 && px < a.refl.fw && py < a.refl.fh;
    const uint32_t c = load_u32_clamped(a.refl, px, py);
    stage_refl[k] = in_refl ? c : 0u;
    lit |= stage_refl[k] & 0x00FFFFFFu;  // alpha is not read by the taps
  }
  // the sRGB table is staged here, behind the loads above, so that its own load is waited for together with them (its
  // readers come after the barriers below)
  srgb_lut_stage(s_lut, tid, BLUR_THREADS);
  const bool empty_tile = a.skip_empty_tiles != 0 && __syncthreads_or(lit != 0u) == 0;
  if (!empty_tile) {
    BilinearTaps stage_normal[STAGE_ITERS];
    uint32_t stage_depth[STAGE_ITERS];
#pragma unroll
    for (int k = 0; k < STAGE_ITERS; k++) {
      const int t = min(tid + k * BLUR_THREADS, BLUR_TW * BLUR_TH - 1);
      const int tx = t % BLUR_TW, ty = t / BLUR_TW;
      const int px = bx0 + tx, py = by0 + ty;
      const f2 uv = mk2((float)px / tex_size.x, (float)py / tex_size.y);
      stage_normal[k] = bilinear_taps_u32(a.normal, uv);
      const bool in_depth = px >= 0 && py >= 0 && px < a.depth1.fw && py < a.depth1.fh;
      const uint32_t d = load_u32_clamped(a.depth1, px, py);
      stage_depth[k] = in_depth ? d : 0u;  // D24 word 0 decodes to 0.0f
    }
#pragma unroll
    for (int k = 0; k < STAGE_ITERS; k++) {
      const int t = min(tid + k * BLUR_THREADS, BLUR_TW * BLUR_TH - 1);
      const f3 n = decode_normal_fast(taps_resolve<FmtRG16U>(stage_normal[k]));  // only enters the normal weight
      lds_store4(&s_px[t], blur_pack(n, FmtD24::decode(stage_depth[k]), stage_refl[k]));
    }
  }

  // (3): prev_uv needs the velocity, which has arrived by now
  f2 velocity[2], prev_uv[2];
  BilinearTaps rough_taps[2], normal_taps[2], depth_taps[2], prev_depth_taps[2], history_taps[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    velocity[k] = taps_resolve<FmtRG16F>(velocity_taps[k]);
    prev_uv[k] = uv_c[k] + velocity[k];
    const f2 safe_prev = mk2(vclamp(prev_uv[k].x, 0.0f, 1.0f), vclamp(prev_uv[k].y, 0.0f, 1.0f));  // only consumed when prev_uv is inside
    if (!empty_tile) {
      rough_taps[k] = bilinear_taps_u32(a.material, uv_c[k]);
      normal_taps[k] = bilinear_taps_u32(a.normal, uv_c[k]);
    }
    depth_taps[k] = bilinear_taps_u32(a.depth1, uv_c[k]);
    prev_depth_taps[k] = bilinear_taps_u32(a.hist_depth1, safe_prev);
    history_taps[k] = bilinear_taps_u32(a.history, uv_c[k]);  // screen_uv, not prev_uv (blur.comp:103)
  }
  __syncthreads();
  VKR_STAMP(1);  // tile staged, per-pixel loads issued
  if (!live) return;

  // per-pixel set-up (blur.comp:34-55) and the temporal test (blur.comp:79-105) for A and B
  BlurCentre c[2];
  bool reprojected[2];
  f3 history_color[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (!empty_tile) {
      float roughness = taps_srgb_channel(rough_taps[k], 1, s_lut);
      roughness = mixf(0.0f, a.max_roughness, roughness);
      c[k].tc = (2 * threadIdx.y + k + BLUR_R) * BLUR_TW + (threadIdx.x + BLUR_R);
      c[k].depth = __uint_as_float(s_px[c[k].tc].w);
      c[k].normal = decode_normal_fast(taps_resolve<FmtRG16U>(normal_taps[k]));
      float sigma = mixf(0.4f, 4.0f, roughness);
      if (a.disable_blur != 0) sigma = 0.35f;
      c[k].r = min(f2i(floorf(3.0f * sigma - 0.01f)), BLUR_R);
      c[k].g = 1.0f / (((2.0f * VKR_PI) * sigma) * sigma);
      const float e = (2.0f * sigma) * sigma;
      c[k].neg_inv_e_log2 = -1.4426950408889634f / e;
      c[k].k_bilateral = 1000.0f / c[k].depth;
    }

    reprojected[k] = false;
    const f2 screen_uv = uv_c[k];
    const float delta_len = length(velocity[k]);
    if (prev_uv[k].x >= 0.0f && prev_uv[k].y >= 0.0f && prev_uv[k].x <= 1.0f && prev_uv[k].y <= 1.0f) {
      const f3 vc = reconstruct_view_vec(screen_uv, taps_resolve<FmtD24>(depth_taps[k]), a.pr);
      const f3 v_world_cur = xyz(mul(a.inverse_camera, mk4(vc.x, vc.y, vc.z, 1.0f)));
      const f3 vp = reconstruct_view_vec(prev_uv[k], taps_resolve<FmtD24>(prev_depth_taps[k]), a.pr);
      const f3 v_world_prev = xyz(mul(a.prev_inverse_camera, mk4(vp.x, vp.y, vp.z, 1.0f)));
      const f3 v_camera = xyz(mul(a.inverse_camera, mk4(0, 0, 0, 1)));
      const float error = length(v_world_cur - v_world_prev);
      const float pixel_dist = length(v_world_cur - v_camera);
      reprojected[k] = (delta_len < 0.0001f) || (error < vclamp((0.1f * pixel_dist) * delta_len, 0.01f, 0.1f));
    }
    if (a.accumulate == 0) reprojected[k] = false;
    history_color[k] = taps_resolve<FmtRGBA8>(history_taps[k]);
  }

  VKR_STAMP(2);  // per-pixel set-up done
  VKR_STAMP_VALUE(5, empty_tile ? 0 : c[0].r);
  f4 accA = mk4(0, 0, 0, 0), accB = mk4(0, 0, 0, 0);
  bool uniform_sigma = false;
  if (!empty_tile && a.uniform_sigma_path != 0) {
    const uint32_t key = __float_as_uint(c[0].neg_inv_e_log2);
    const uint32_t key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    const bool same = has_b && key == key0 && __float_as_uint(c[1].neg_inv_e_log2) == key0;
    if (__ballot(same) == ~0ull) {  // a full wave, every pixel with the same sigma (hence the same radius and normalisation)
      uniform_sigma = true;
      switch (__builtin_amdgcn_readfirstlane(c[0].r)) {
        case 1: blur_uniform_sigma<1>(s_px, c, accA, accB); break;
        case 2: blur_uniform_sigma<2>(s_px, c, accA, accB); break;
        case 3: blur_uniform_sigma<3>(s_px, c, accA, accB); break;
        case 4: blur_uniform_sigma<4>(s_px, c, accA, accB); break;
        case 5: blur_uniform_sigma<5>(s_px, c, accA, accB); break;
        case 6: blur_uniform_sigma<6>(s_px, c, accA, accB); break;
        case 7: blur_uniform_sigma<7>(s_px, c, accA, accB); break;
        case 8: blur_uniform_sigma<8>(s_px, c, accA, accB); break;
        case 9: blur_uniform_sigma<9>(s_px, c, accA, accB); break;
        case 10: blur_uniform_sigma<10>(s_px, c, accA, accB); break;
        // (radius 11 — roughness near 1 — is left to blur_rows<11>: measured faster than blur_uniform_sigma<11>, whose 12 factor pairs
        // no longer fit beside the unrolled rows: 0.724 against 0.768 ms per frame with the empty-tile skips off, where the sky's tiles count)
        default: uniform_sigma = false; break;
      }
    }
  }
  bool by_rows = false;
  if (!empty_tile && !uniform_sigma && a.rows_path != 0) {
    // the largest radius of the wave (lanes past the window's bottom row have no pixel B: its radius does not count)
    int rmax = max(c[0].r, has_b ? c[1].r : 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rmax = max(rmax, __shfl_xor(rmax, off));
    by_rows = true;
    switch (__builtin_amdgcn_readfirstlane(rmax)) {
      case 0: case 1: blur_rows<1>(s_px, c, has_b, accA, accB); break;
      case 2: blur_rows<2>(s_px, c, has_b, accA, accB); break;
      case 3: blur_rows<3>(s_px, c, has_b, accA, accB); break;
      case 4: blur_rows<4>(s_px, c, has_b, accA, accB); break;
      case 5: blur_rows<5>(s_px, c, has_b, accA, accB); break;
      case 6: blur_rows<6>(s_px, c, has_b, accA, accB); break;
      case 7: blur_rows<7>(s_px, c, has_b, accA, accB); break;
      case 8: blur_rows<8>(s_px, c, has_b, accA, accB); break;
      case 9: blur_rows<9>(s_px, c, has_b, accA, accB); break;
      case 10: blur_rows<10>(s_px, c, has_b, accA, accB); break;
      case 11: blur_rows<11>(s_px, c, has_b, accA, accB); break;
      default: by_rows = false; break;
    }
  }
  if (empty_tile || uniform_sigma || by_rows) {
    // empty: nothing to add up, both sums stay 0 and the epilogue turns them into colour 0 (then the history mix)
  } else if (!has_b || c[0].r != c[1].r) {
    accA = blur_single(s_px, c[0]);
    if (has_b) accB = blur_single(s_px, c[1]);
  } else {
    // paired path: lane x = pixel A, lane y = pixel B, same radius r.  A's rows are jA = -r..r,
    // B's rows jB = jA - 1; the staged row tcA + jA*TW serves both for jA = -r+1..r.
    const int r = c[0].r;
    const int tcA = c[0].tc;
    const v2f cnx = {c[0].normal.x, c[1].normal.x}, cny = {c[0].normal.y, c[1].normal.y}, cnz = {c[0].normal.z, c[1].normal.z};
    const v2f cd = {c[0].depth, c[1].depth}, kb = {c[0].k_bilateral, c[1].k_bilateral}, g2 = {c[0].g, c[1].g};
    const v2f nie = {c[0].neg_inv_e_log2, c[1].neg_inv_e_log2};
    const v2f kappa = exp2_2(2.0f * nie);
    const v2f e_edge = exp2_2((float)(r * r) * nie);             // E(+-r)
    const v2f rho_edge = exp2_2((float)(1 - 2 * r) * nie);       // rho(-r)
    // first paired row jA = -r+1: lane A sits at E(-r+1), lane B at E(-r)
    const v2f ej_first = {e_edge.x * rho_edge.x, e_edge.y};
    const v2f rhoj_first = {rho_edge.x * kappa.x, rho_edge.y};
    v2f acc_r = {0.0f, 0.0f}, acc_g = {0.0f, 0.0f}, acc_b = {0.0f, 0.0f}, acc_w = {0.0f, 0.0f};
    f4 edgeA = mk4(0, 0, 0, 0), edgeB = mk4(0, 0, 0, 0);
    v2f ei = e_edge, rho_i = rho_edge;
    const v2f kcd = kb * cd;  // bilateral term: 1 - min(|k*cd - k*d|, 1)
    // The normalisation g is applied once after the loop, so every running weight E(i)*E(j)*nw*bil
    // stays in [0, 1] and the two max(., 0) of the shader become the free `clamp` output modifier.
#pragma unroll 1
    for (int i = -r; i <= r; i++) {
      const int col = tcA + i;
      // unpaired ends of the column: A's row -r and B's row +r (tcB + r*TW = tcA + (r+1)*TW)
      blur_tap(s_px, c[0], col - r * BLUR_TW, ei.x * e_edge.x, edgeA);
      blur_tap(s_px, c[1], col + (r + 1) * BLUR_TW, ei.y * e_edge.y, edgeB);
      v2f wj = ei * ej_first, rho_j = rhoj_first;  // running E(i) * E(j)
      int t = col - (r - 1) * BLUR_TW;
#pragma unroll 2
      for (int j = -r + 1; j <= r; j++, t += BLUR_TW) {
        const uint4 p = lds_load4(&s_px[t]);
        const v2f nzw = {__uint_as_float(p.z), __uint_as_float(p.w)};
        const v2f nxy = __builtin_elementwise_fma(cny, splat2(__uint_as_float(p.y)), cnx * splat2(__uint_as_float(p.x)));
        v2f nw, kdz;
        // nw = clamp(cnz * n.z + nxy): v_pk_* has no max, the clamp modifier does it for free
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(nw) : "v"(cnz), "v"(nzw), "v"(nxy));
        // kdz = kcd - kb * n.w (both lanes read the high half of {n.z, n.w})
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
            : "=v"(kdz) : "v"(kb), "v"(nzw), "v"(kcd));
        const v2f tw = wj * nw;
        // tw * max(1 - |kdz|, 0) = clamp(tw - |kdz| * tw) since 0 <= tw <= 1: one v_fma_f32 with |.| and clamp
        v2f w;
        w.x = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.x), tw.x, tw.x), 0.0f), 1.0f);
        w.y = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-__builtin_fabsf(kdz.y), tw.y, tw.y), 0.0f), 1.0f);
        acc_r = __builtin_elementwise_fma(w, splat2((float)(p.x & 0xFFu)), acc_r);
        acc_g = __builtin_elementwise_fma(w, splat2((float)(p.y & 0xFFu)), acc_g);
        acc_b = __builtin_elementwise_fma(w, splat2((float)(p.z & 0xFFu)), acc_b);
        acc_w += w;
        wj *= rho_j; rho_j *= kappa;
      }
      ei *= rho_i; rho_i *= kappa;
    }
    accA = mk4(acc_r.x + edgeA.x, acc_g.x + edgeA.y, acc_b.x + edgeA.z, acc_w.x + edgeA.w);
    accA = mk4(accA.x * g2.x, accA.y * g2.x, accA.z * g2.x, accA.w * g2.x);
    accB = mk4(acc_r.y + edgeB.x, acc_g.y + edgeB.y, acc_b.y + edgeB.z, acc_w.y + edgeB.w);
    accB = mk4(accB.x * g2.y, accB.y * g2.y, accB.z * g2.y, accB.w * g2.y);
  }

  VKR_STAMP(3);  // tap loop done (wave 0)
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (k == 1 && !has_b) break;
    const f4 acc = k == 0 ? accA : accB;
    f3 color = mk3(acc.x, acc.y, acc.z) * (1.0f / 255.0f);
    color = color / vmax(acc.w, 0.001f);
    if (reprojected[k]) color = mix3(history_color[k], color, 0.1f);
    *texel_ptr<uint32_t>(a.out, lx, lyA + k) =
        float_to_unorm8(color.x) | (float_to_unorm8(color.y) << 8) | (float_to_unorm8(color.z) << 16);
  }
}

__global__ __launch_bounds__(BLUR_THREADS, BLUR_WAVES) void k_sssr_blur(BlurArgs a) {
  __shared__ uint4 s_px[BLUR_TH * BLUR_TW];
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  const int tid = threadIdx.y * BLUR_BX + threadIdx.x;
  VKR_STAMP(0);
#ifdef VKR_BLUR_STAMPS
  {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    VKR_STAMP_VALUE(6, ((unsigned long long)xcc << 32) | hw);
  }
#endif
  blur_tile(a, xcd_block<4, 64 / BLUR_BY>(), s_px, s_lut, tid);  // chunks of 128 x 64 output pixels; stages s_lut
  VKR_STAMP(4);
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_pdf_preintegrate(const vkr_img* out_pdf, void* stream) {
  Tex out;
  VKR_TRY(make_tex(out_pdf, 0, VKR_FMT_R32_SFLOAT, "pdf_preintegrate.out", &out));
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_pdf_preintegrate, grid2d(out.w, out.h, block), block, 0, (hipStream_t)stream, out);
  return launch_status("pdf_preintegrate");
}

static void load_proj(Proj& pr, float fovy, float aspect, float znear, float zfar) {
  pr.tg = tanf(fovy / 2.0f);
  pr.aspect = aspect; pr.znear = znear; pr.zfar = zfar;
}

static int make_trace_args(TraceArgs& a, const vkr_img* depth, const vkr_img* normal, const vkr_img* material, const vkr_trace_params* params,
                           const float* halton_vec4, const vkr_img* out_ray, const vkr_img* out_occlusion, const vkr_img* pdf_tex,
                           const vkr_trace_push* push) {
  if (!params || !push || !halton_vec4 || !depth) { set_error("sssr_trace: NULL argument"); return VKR_ERR_NULL; }
  if (((uintptr_t)halton_vec4 % 16) != 0) { set_error("sssr_trace: halton buffer must be 16-byte aligned"); return VKR_ERR_LAYOUT; }
  if (depth->mip_count < 1 || depth->mip_count > VKR_MAX_MIPS) { set_error("sssr_trace: bad depth mip count"); return VKR_ERR_MIPS; }
  a.depth.count = (int)depth->mip_count;
  for (int i = 0; i < a.depth.count; i++) VKR_TRY(make_tex(depth, i, VKR_FMT_D24_UNORM_S8, "sssr_trace.depth", &a.depth.mip[i]));
  for (int i = a.depth.count; i < 16; i++) a.depth.mip[i] = a.depth.mip[0];
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "sssr_trace.normal", &a.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "sssr_trace.material", &a.material));
  VKR_TRY(make_tex(pdf_tex, 0, VKR_FMT_R32_SFLOAT, "sssr_trace.pdf", &a.pdf));
  VKR_TRY(make_tex(out_ray, 0, VKR_FMT_RGBA16_UNORM, "sssr_trace.out_ray", &a.out_ray));
  VKR_TRY(make_tex(out_occlusion, 0, VKR_FMT_RGBA16_SFLOAT, "sssr_trace.out_occlusion", &a.out_occ));
  if (!same_window(a.out_ray, a.out_occ)) { set_error("sssr_trace: ray / occlusion outputs differ in extent"); return VKR_ERR_EXTENT; }
  for (int i = 0; i < a.depth.count; i++) {
    const Tex& m = a.depth.mip[i];
    if (m.ox != 0 || m.oy != 0 || m.w != m.fw || m.h != m.fh || m.w > 65535 || m.h > 65535) {
      set_error("sssr_trace: the Hi-Z pyramid must cover the whole frame (rays have unbounded reach)");
      return VKR_ERR_EXTENT;
    }
  }
  a.halton = (const float4*)halton_vec4;
  load_mat(a.normal_mat, params->normal_mat);
  load_proj(a.pr, params->fovy, params->aspect, params->znear, params->zfar);
  a.frame_random = params->frame_random;
  a.max_roughness = push->max_roughness;
  a.horizon_d2 = horizon_threshold_d2();
  {
    const float sw = (float)a.depth.mip[0].fw, sh = (float)a.depth.mip[0].fh;
    a.screen_size_inv.x = 1.0f / sw; a.screen_size_inv.y = 1.0f / sh;
    a.uv_offset_abs.x = 0.005f / sw; a.uv_offset_abs.y = 0.005f / sh;
    a.f_over_fn = a.pr.zfar / (a.pr.zfar - a.pr.znear);
  }
  a.nrm_row0 = 0; a.nrm_row1 = a.normal.fh;
  a.pend_mask = a.out_ray; a.pend_data = a.out_ray;  // unused unless windowed
  a.q.records = nullptr; a.q.counters = nullptr; a.q.capacity = 0; a.park_after = 0;
  a.local.count = 0;
  for (int i = 0; i < 16; i++) a.local.mip[i] = a.depth.mip[0];
  return VKR_OK;
}

extern "C" int vkr_sssr_trace(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                              const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                              const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_trace_push* push,
                              void* stream) {
  TraceArgs a;
  VKR_TRY(make_trace_args(a, depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, push));
  dim3 block(TRACE_THREADS, 1);
  dim3 grid((a.out_ray.w + 31) / 32, (a.out_ray.h + 8 * TRACE_WY - 1) / (8 * TRACE_WY));
  hipLaunchKernelGGL((k_sssr_trace<false, false, false>), grid, block, 0, (hipStream_t)stream, a);
  return launch_status("sssr_trace");
}

// ---- sssr_trace in two launches (include/vkr_postfx.h: vkr_sssr_trace_split) --------------------------------------------
#define TQ_HEADER_BYTES 256u
extern "C" uint64_t vkr_sssr_trace_workspace_bytes(uint32_t rays_width, uint32_t rays_height) {
  return (uint64_t)TQ_HEADER_BYTES + (uint64_t)rays_width * rays_height * (TQ_VEC * sizeof(uint4));
}

static int bind_trace_queue(TraceArgs& a, void* workspace, uint64_t workspace_bytes, const char* what) {
  if (!workspace || ((uintptr_t)workspace % 16) != 0) { set_error("%s: workspace NULL or not 16-byte aligned", what); return VKR_ERR_NULL; }
  const uint64_t need = vkr_sssr_trace_workspace_bytes((uint32_t)a.out_ray.w, (uint32_t)a.out_ray.h);
  if (workspace_bytes < need) { set_error("%s: workspace of %llu bytes, %llu needed (vkr_sssr_trace_workspace_bytes)", what, (unsigned long long)workspace_bytes, (unsigned long long)need); return VKR_ERR_EXTENT; }
  if (a.out_ray.w > 65535 || a.out_ray.h > 65535) { set_error("%s: a parked ray names its pixel in 16 + 16 bits", what); return VKR_ERR_EXTENT; }
  a.q.counters = (uint32_t*)workspace;
  a.q.records = (uint4*)((uint8_t*)workspace + TQ_HEADER_BYTES);
  a.q.capacity = (uint32_t)a.out_ray.w * (uint32_t)a.out_ray.h;
  return VKR_OK;
}

// blocks of the resume launch: they loop over the queue, so the grid only has to fill the machine
static dim3 resume_grid(const TraceArgs& a) {
  const uint32_t chunks = (a.q.capacity + RESUME_BLOCK - 1) / RESUME_BLOCK;
  const uint32_t most = 2048u * (TRACE_THREADS / RESUME_BLOCK);
  return dim3(chunks < most ? chunks : most);
}

extern "C" int vkr_sssr_trace_split(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                    const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                    const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_trace_push* push,
                                    void* workspace, uint64_t workspace_bytes, uint32_t park_after_rounds, void* stream) {
  TraceArgs a;
  VKR_TRY(make_trace_args(a, depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, push));
  VKR_TRY(bind_trace_queue(a, workspace, workspace_bytes, "sssr_trace_split"));
  if (park_after_rounds > 4) { set_error("sssr_trace_split: park_after_rounds %u (0..4: a march has at most four compacted rounds)", park_after_rounds); return VKR_ERR_EXTENT; }
  a.park_after = (int)park_after_rounds;
  dim3 block(TRACE_THREADS, 1);
  dim3 grid((a.out_ray.w + 31) / 32, (a.out_ray.h + 8 * TRACE_WY - 1) / (8 * TRACE_WY));
  // the queue's counter: zeroed on the stream ahead of the head launch (a reset by the resume launch itself needs every one of
  // its blocks to report that it has read the count: 2048 atomics on one word, 11 ns each, measured as 22 us)
  if (hipMemsetAsync(a.q.counters, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) { set_error("sssr_trace_split: memset failed"); return VKR_ERR_LAYOUT; }
  hipLaunchKernelGGL((k_sssr_trace<false, true, false>), grid, block, 0, (hipStream_t)stream, a);
  VKR_TRY(launch_status("sssr_trace_split (head)"));
  hipLaunchKernelGGL((k_sssr_trace_resume<false, false>), resume_grid(a), dim3(RESUME_BLOCK), 0, (hipStream_t)stream, a);
  return launch_status("sssr_trace_split (resume)");
}

static int make_windowed_args(TraceArgs& a, const vkr_img* depth, const vkr_img* normal, const vkr_img* material, const vkr_trace_params* params,
                              const float* halton_vec4, const vkr_img* out_ray, const vkr_img* out_occlusion, const vkr_img* pdf_tex,
                              const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_trace_window_push* push, const char* what) {
  if (!push) { set_error("%s: NULL argument", what); return VKR_ERR_NULL; }
  const vkr_trace_push base {push->max_roughness};
  VKR_TRY(make_trace_args(a, depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, &base));
  if (a.normal.ox != 0 || a.normal.oy != 0 || a.normal.w != a.normal.fw || a.normal.h != a.normal.fh) {
    set_error("%s: `normal` must be the whole-frame image (rows outside the window are filled later)", what);
    return VKR_ERR_EXTENT;
  }
  if (push->normal_row0 >= push->normal_row1 || push->normal_row1 > (uint32_t)a.normal.fh) {
    set_error("%s: normal rows [%u, %u) of %d", what, push->normal_row0, push->normal_row1, a.normal.fh);
    return VKR_ERR_EXTENT;
  }
  a.nrm_row0 = (int)push->normal_row0; a.nrm_row1 = (int)push->normal_row1;
  VKR_TRY(make_tex(pending_mask, 0, VKR_FMT_R8_UNORM, "sssr_trace_windowed.pending_mask", &a.pend_mask));
  VKR_TRY(make_tex(pending_data, 0, VKR_FMT_RGBA32_SFLOAT, "sssr_trace_windowed.pending_data", &a.pend_data));
  if (a.pend_mask.w != a.out_ray.w || a.pend_mask.h != a.out_ray.h || a.pend_data.w != 2 * a.out_ray.w || a.pend_data.h != a.out_ray.h) {
    set_error("%s: pending_mask must have the rays' extent, pending_data twice its width", what);
    return VKR_ERR_EXTENT;
  }
  return VKR_OK;
}

extern "C" int vkr_sssr_trace_windowed(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                       const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                       const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                       const vkr_img* pending_data, const vkr_trace_window_push* push, void* stream) {
  TraceArgs a;
  VKR_TRY(make_windowed_args(a, depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, pending_mask, pending_data, push, "sssr_trace_windowed"));
  dim3 block(TRACE_THREADS, 1);
  dim3 grid((a.out_ray.w + 31) / 32, (a.out_ray.h + 8 * TRACE_WY - 1) / (8 * TRACE_WY));
  hipLaunchKernelGGL((k_sssr_trace<true, false, false>), grid, block, 0, (hipStream_t)stream, a);
  return launch_status("sssr_trace_windowed");
}

// The windowed trace in two launches around the arrival of the whole-frame pyramid (include/vkr_postfx.h).
extern "C" int vkr_sssr_trace_windowed_head(const vkr_img* local_depth, const vkr_img* frame_depth, const vkr_img* normal, const vkr_img* material,
                                            const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                            const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                            const vkr_img* pending_data, const vkr_trace_window_push* push, void* workspace,
                                            uint64_t workspace_bytes, uint32_t park_after_rounds, void* stream) {
  const char* P = "sssr_trace_windowed_head";
  TraceArgs a;
  VKR_TRY(make_windowed_args(a, frame_depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, pending_mask, pending_data, push, P));
  VKR_TRY(bind_trace_queue(a, workspace, workspace_bytes, P));
  if (park_after_rounds > 4) { set_error("%s: park_after_rounds %u (0..4)", P, park_after_rounds); return VKR_ERR_EXTENT; }
  a.park_after = (int)park_after_rounds;
  if (!local_depth || local_depth->mip_count < 1 || local_depth->mip_count > (uint32_t)a.depth.count) { set_error("%s: the local pyramid must have 1 .. %d levels", P, a.depth.count); return VKR_ERR_MIPS; }
  a.local.count = (int)local_depth->mip_count;
  // every local level must be made of whole texels of the frame's level: make_tex() halves origin and extent per level, which
  // is only the frame's texel grid when both divide (a strip cut at a multiple of 16 full-res rows has four such levels)
  if (((uint32_t)local_depth->origin_y | local_depth->height) & ((1u << (a.local.count - 1)) - 1u)) {
    set_error("%s: local rows [%d, +%u) do not divide into %d levels (origin and height must be multiples of %u)", P, local_depth->origin_y,
              local_depth->height, a.local.count, 1u << (a.local.count - 1));
    return VKR_ERR_EXTENT;
  }
  for (int i = 0; i < a.local.count; i++) {
    Tex& t = a.local.mip[i];
    VKR_TRY(make_tex(local_depth, i, VKR_FMT_D24_UNORM_S8, "sssr_trace_windowed_head.local_depth", &t));
    const Tex& f = a.depth.mip[i];
    // a strip of the frame's level: full width, rows [oy, oy + h) — and the same texel grid as the frame's level
    if (t.ox != 0 || t.w != t.fw || t.fw != f.fw || t.fh != f.fh) {
      set_error("%s: level %d of the local pyramid (%dx%d of %dx%d at row %d) is not a strip of the frame's level (%dx%d)", P, i, t.w, t.h, t.fw, t.fh, t.oy, f.fw, f.fh);
      return VKR_ERR_EXTENT;
    }
  }
  for (int i = a.local.count; i < 16; i++) a.local.mip[i] = a.local.mip[0];
  // the pixels of this launch sample level 0 at their own position: their rows must be among the local ones
  if (a.out_ray.oy < a.local.mip[0].oy || a.out_ray.oy + a.out_ray.h > a.local.mip[0].oy + a.local.mip[0].h) {
    set_error("%s: the rays' rows [%d, %d) are not inside the local rows of level 0 [%d, %d)", P, a.out_ray.oy, a.out_ray.oy + a.out_ray.h,
              a.local.mip[0].oy, a.local.mip[0].oy + a.local.mip[0].h);
    return VKR_ERR_EXTENT;
  }
  if (hipMemsetAsync(a.q.counters, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) { set_error("%s: memset failed", P); return VKR_ERR_LAYOUT; }
  dim3 block(TRACE_THREADS, 1);
  dim3 grid((a.out_ray.w + 31) / 32, (a.out_ray.h + 8 * TRACE_WY - 1) / (8 * TRACE_WY));
  hipLaunchKernelGGL((k_sssr_trace<true, true, true>), grid, block, 0, (hipStream_t)stream, a);
  return launch_status(P);
}

extern "C" int vkr_sssr_trace_windowed_resume(const vkr_img* frame_depth, const vkr_img* normal, const vkr_img* material,
                                              const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                              const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                              const vkr_img* pending_data, const vkr_trace_window_push* push, void* workspace,
                                              uint64_t workspace_bytes, void* stream) {
  const char* P = "sssr_trace_windowed_resume";
  TraceArgs a;
  VKR_TRY(make_windowed_args(a, frame_depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, pending_mask, pending_data, push, P));
  VKR_TRY(bind_trace_queue(a, workspace, workspace_bytes, P));
  hipLaunchKernelGGL((k_sssr_trace_resume<true, true>), resume_grid(a), dim3(RESUME_BLOCK), 0, (hipStream_t)stream, a);
  return launch_status(P);
}

extern "C" int vkr_sssr_validate(const vkr_img* rays, const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_img* frame_normals,
                                 const vkr_trace_params* params, void* stream) {
  return vkr_sssr_validate_unless(rays, pending_mask, pending_data, frame_normals, params, nullptr, stream);
}

extern "C" int vkr_sssr_validate_unless(const vkr_img* rays, const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_img* frame_normals,
                                        const vkr_trace_params* params, const uint32_t* skip_if_set, void* stream) {
  if (!params) { set_error("sssr_validate: NULL argument"); return VKR_ERR_NULL; }
  Tex r, m, pd, n;
  VKR_TRY(make_tex(rays, 0, VKR_FMT_RGBA16_UNORM, "sssr_validate.rays", &r));
  VKR_TRY(make_tex(pending_mask, 0, VKR_FMT_R8_UNORM, "sssr_validate.pending_mask", &m));
  VKR_TRY(make_tex(pending_data, 0, VKR_FMT_RGBA32_SFLOAT, "sssr_validate.pending_data", &pd));
  VKR_TRY(make_tex(frame_normals, 0, VKR_FMT_RG16_UNORM, "sssr_validate.normal", &n));
  if (m.w != r.w || m.h != r.h || pd.w != 2 * r.w || pd.h != r.h) { set_error("sssr_validate: pending images do not match the rays"); return VKR_ERR_EXTENT; }
  Mat4 nm;
  load_mat(nm, params->normal_mat);
  const dim3 block(64, 4);
  hipLaunchKernelGGL(k_sssr_validate, grid2d(r.w, r.h, block), block, 0, (hipStream_t)stream, r, m, pd, n, nm, skip_if_set);
  return launch_status("sssr_validate");
}

extern "C" int vkr_sssr_filter(const vkr_img* rays, const vkr_img* depth, const vkr_img* albedo, const vkr_img* normal,
                               const vkr_img* material, const vkr_img* out_reflections, const vkr_trace_params* params,
                               const vkr_filter_push* push, void* stream) {
  if (!params || !push) { set_error("sssr_filter: NULL argument"); return VKR_ERR_NULL; }
  FilterArgs a;
  VKR_TRY(make_tex(rays, 0, VKR_FMT_RGBA16_UNORM, "sssr_filter.rays", &a.rays));
  VKR_TRY(make_tex(depth, 1, VKR_FMT_D24_UNORM_S8, "sssr_filter.depth (level 1)", &a.depth1));  // filter.comp:75,121
  VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "sssr_filter.albedo", &a.albedo));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "sssr_filter.normal", &a.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "sssr_filter.material", &a.material));
  VKR_TRY(make_tex(out_reflections, 0, VKR_FMT_RGBA8_UNORM, "sssr_filter.out", &a.out));
  load_mat(a.normal_mat, params->normal_mat);
  load_proj(a.pr, params->fovy, params->aspect, params->znear, params->zfar);
  a.render_flags = push->render_flags;
  a.skip_empty_tiles = (switches() & VKR_SWITCH_FILTER_NO_SKIP) ? 0u : 1u;  // measurement switch: identical output for finite weights (see the kernel)
  dim3 block(FILT_BX, FILT_BY);
  hipLaunchKernelGGL(k_sssr_filter, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("sssr_filter");
}

static int make_blur_args(BlurArgs& a, const vkr_img* depth, const vkr_img* normal, const vkr_img* reflections, const vkr_img* material,
                          const vkr_img* history, const vkr_img* velocity, const vkr_img* history_depth, const vkr_img* out_blurred,
                          const vkr_reproject_params* params, const vkr_blur_push* push) {
  if (!params || !push) { set_error("sssr_blur: NULL argument"); return VKR_ERR_NULL; }
  VKR_TRY(make_tex(depth, 1, VKR_FMT_D24_UNORM_S8, "sssr_blur.depth (level 1)", &a.depth1));  // blur.comp:42,62,111
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "sssr_blur.normal", &a.normal));
  VKR_TRY(make_tex(reflections, 0, VKR_FMT_RGBA8_UNORM, "sssr_blur.reflections", &a.refl));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "sssr_blur.material", &a.material));
  VKR_TRY(make_tex(history, 0, VKR_FMT_RGBA8_UNORM, "sssr_blur.history", &a.history));
  VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "sssr_blur.velocity", &a.velocity));
  VKR_TRY(make_tex(history_depth, 1, VKR_FMT_D24_UNORM_S8, "sssr_blur.history_depth (level 1)", &a.hist_depth1));
  VKR_TRY(make_tex(out_blurred, 0, VKR_FMT_RGBA8_UNORM, "sssr_blur.out", &a.out));
  load_mat(a.inverse_camera, params->inverse_camera);
  load_mat(a.prev_inverse_camera, params->prev_inverse_camera);
  load_proj(a.pr, params->fovy_aspect_znear_zfar[0], params->fovy_aspect_znear_zfar[1], params->fovy_aspect_znear_zfar[2],
            params->fovy_aspect_znear_zfar[3]);
  a.max_roughness = push->max_roughness;
  a.accumulate = push->accumulate;
  a.disable_blur = push->disable_blur;
  a.skip_empty_tiles = (switches() & VKR_SWITCH_BLUR_NO_SKIP) ? 0u : 1u;  // measurement switch: identical output for finite weights
  a.uniform_sigma_path = (switches() & VKR_SWITCH_BLUR_GENERIC) ? 0u : 1u;
  a.rows_path = (switches() & VKR_SWITCH_BLUR_LANE_LOOPS) ? 0u : 1u;
  if (a.max_roughness > 1.0f || a.max_roughness < 0.0f) {  // sigma <= 4 bounds the staged radius (blur.comp:45)
    set_error("sssr_blur: max_roughness must be in [0,1] (reference slider range, advanced_ssr.cpp:558)");
    return VKR_ERR_EXTENT;
  }
  return VKR_OK;
}

#ifdef VKR_BLUR_STAMPS
extern "C" int vkr_debug_blur_stamps(void* dst, uint64_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_blur_stamps), bytes < sizeof(g_blur_stamps) ? bytes : sizeof(g_blur_stamps));
}
#endif

extern "C" int vkr_sssr_blur(const vkr_img* depth, const vkr_img* normal, const vkr_img* reflections,
                             const vkr_img* material, const vkr_img* history, const vkr_img* velocity,
                             const vkr_img* history_depth, const vkr_img* out_blurred,
                             const vkr_reproject_params* params, const vkr_blur_push* push, void* stream) {
  BlurArgs a;
  VKR_TRY(make_blur_args(a, depth, normal, reflections, material, history, velocity, history_depth, out_blurred, params, push));
  dim3 block(BLUR_BX, BLUR_BY / 2);  // each thread resolves two vertically adjacent pixels
  dim3 grid((a.out.w + BLUR_BX - 1) / BLUR_BX, (a.out.h + BLUR_BY - 1) / BLUR_BY);
  hipLaunchKernelGGL(k_sssr_blur, grid, block, 0, (hipStream_t)stream, a);
  return launch_status("sssr_blur");
}

