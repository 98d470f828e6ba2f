// frame.cpp — see frame.hpp.  C++ exceptions thrown by the pass / graph code (the reference's
// only error channel, gpu/common.cpp:6-12) are turned into status codes at this C boundary.
#include "frame.hpp"
#include "gpu_transfer.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "advanced_ssr.hpp"
#include "defered_shading.hpp"
#include "downsample_pass.hpp"
#include "gtao.hpp"
#include "image_readback.hpp"
#include "scene_renderer.hpp"
#include "screen_trace.hpp"
#include "synthetic_gbuffer.hpp"
#include "taa.hpp"

namespace {

thread_local std::string g_error;

struct GraphInit {  // orders construction: the frame window must be set before any pass creates images
  GraphInit(rendergraph::RenderGraph& g, const vkrh_config& c) {
    g.set_frame_window(c.full_width, c.full_height, c.origin_x, c.origin_y, c.width, c.height);
  }
};

struct PostFxFrame {
  vkrh_config cfg;
  rendergraph::RenderGraph graph;
  GraphInit init;
  struct TransferInit { explicit TransferInit(rendergraph::RenderGraph& g) { gpu_transfer::init(g); } };  // main.cpp: before the passes
  TransferInit transfer_init;
  Gbuffer gbuffer;
  DownsamplePass downsample_pass;
  GTAO gtao;
  AdvancedSSR ssr;
  TAA taa_pass;
  DeferedShadingPass shading_pass;
  rendergraph::ImageResourceId color_out_tex;
  rendergraph::ImageResourceId shadow_map;  // main.cpp passes its shadow map to the shading pass (binding 5; the shader never reads it)
  SyntheticGbuffer synth;
  ScreenSpaceTrace screen_trace;
  ReadBackSystem readback;
  std::unique_ptr<scene::CompiledScene> loaded_scene;
  std::unique_ptr<SceneRenderer> scene_renderer;

  DrawTAAParams draw_params{};
  glm::mat4 projection, view, prev_view;
  bool has_camera = false;
  std::string task_names, task_lanes;

  explicit PostFxFrame(const vkrh_config& c)
      : cfg{c}, graph{c.stream}, init{graph, c}, transfer_init{graph},
        gbuffer{graph, c.width, c.height},
        gtao{graph, c.width, c.height, false, true},  // main.cpp:265: (graph, W, H, USE_RAY_QUERY = 0, half_res = 1)
        ssr{graph, c.width, c.height},
        taa_pass{graph, c.width, c.height},
        shading_pass{graph, nullptr},
        screen_trace{graph, c.width, c.height} {
    if (c.tiled) gbuffer.enable_tiling(graph, c.full_width, c.full_height);
    shadow_map = graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_D24_UNORM_S8_UINT, VK_IMAGE_ASPECT_DEPTH_BIT, 16, 16},
                                    VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_DEPTH_STENCIL_ATTACHMENT_BIT | VK_IMAGE_USAGE_SAMPLED_BIT);
    // main.cpp:289-291
    color_out_tex = graph.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R8G8B8A8_SRGB, VK_IMAGE_ASPECT_COLOR_BIT, c.width, c.height},
                                       VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT | VK_IMAGE_USAGE_SAMPLED_BIT);
  }

  void set_camera(const vkrh_camera& cam) {
    std::memcpy(&view, cam.view, 64);
    std::memcpy(&prev_view, cam.prev_view, 64);
    std::memcpy(&projection, cam.projection, 64);
    // main.cpp:334-339
    draw_params.prev_mvp = projection * prev_view;
    draw_params.mvp = projection * view;
    draw_params.prev_camera = prev_view;
    draw_params.camera = view;
    draw_params.fovy_aspect_znear_zfar = glm::vec4{cam.fovy, cam.aspect, cam.znear, cam.zfar};
    draw_params.jitter = glm::vec4{0.f, 0.f, 0.f, 0.f};
    has_camera = true;
  }

  void run(uint32_t mask) {
    if (!has_camera && (mask & ~uint32_t(VKRH_STAGE_LUT))) throw std::runtime_error{"vkrh_run: camera not set"};
    const glm::vec4 fazz = draw_params.fovy_aspect_znear_zfar;
    if (mask & VKRH_STAGE_LUT) ssr.preintegrate_pdf(graph);
    if (mask & VKRH_STAGE_BRDF_LUT) ssr.preintegrate_brdf(graph);
    if (mask & VKRH_STAGE_PREV_DEPTH) {
      // what the previous frame left behind: its depth in `prev_depth`, including the Hi-Z mips
      synth.draw_depth(graph, gbuffer.prev_depth, prev_view, draw_params.prev_mvp, fazz);
      downsample_pass.run(graph, gbuffer.normal, gbuffer.velocity_vectors, gbuffer.prev_depth, gbuffer.downsampled_normals,
                          gbuffer.downsampled_velocity_vectors);
    }
    if (mask & VKRH_STAGE_GBUFFER) synth.draw_taa(graph, gbuffer, draw_params);
    if (mask & VKRH_STAGE_RASTER) {  // main.cpp:345
      if (!scene_renderer) throw std::runtime_error{"vkrh_run: VKRH_STAGE_RASTER without a loaded scene"};
      scene_renderer->draw_taa(graph, gbuffer, draw_params);
    }
    if (mask & VKRH_STAGE_DOWNSAMPLE)  // main.cpp:347
      downsample_pass.run(graph, gbuffer.normal, gbuffer.velocity_vectors, gbuffer.depth, gbuffer.downsampled_normals,
                          gbuffer.downsampled_velocity_vectors);
    if (mask & VKRH_STAGE_DOWNSAMPLE_NEXT) {
      if (!gbuffer.pipelined) throw std::runtime_error{"vkrh_run: VKRH_STAGE_DOWNSAMPLE_NEXT without the G-buffer's second set"};
      downsample_pass.run(graph, gbuffer.normal, gbuffer.velocity_vectors, gbuffer.depth_next, gbuffer.downsampled_normals_next,
                          gbuffer.downsampled_velocity_vectors_next);
    }
    if ((mask & VKRH_STAGE_HIZ_TAIL) && gbuffer.tiled) {
      // the launcher gathered whole-frame mips 0..k of frame_hiz; finish the chain locally
      downsample_pass.run_downsample_depth(graph, gbuffer.frame_hiz, hiz_gathered_mips - 1);
    }
    // main.cpp:368-373
    const glm::mat4 normal_mat = glm::transpose(glm::inverse(view));
    const GTAOParams gtao_params{normal_mat, fazz.x, fazz.y, fazz.z, fazz.w};
    const AdvancedSSRParams assr_params{normal_mat, fazz.x, fazz.y, fazz.z, fazz.w};
    if (mask & (VKRH_STAGE_SSR | VKRH_STAGE_SSR_CLASSIFIED)) {                               // main.cpp:375
      ssr.get_settings().use_tile_classification = (mask & VKRH_STAGE_SSR_CLASSIFIED) != 0;
      ssr.run(graph, assr_params, draw_params, gbuffer, gtao.raw);
    }
    if (mask & VKRH_STAGE_SSR_TRACE_HEAD) ssr.run_trace_head(graph, assr_params, gbuffer, gtao.raw, hiz_gathered_mips);
    if (mask & VKRH_STAGE_SSR_TRACE_RESUME) ssr.run_trace_resume(graph, gbuffer, gtao.raw);
    if (mask & VKRH_STAGE_SSR_TRACE) ssr.run_trace(graph, assr_params, gbuffer, gtao.raw);
    if (mask & VKRH_STAGE_SSR_RESOLVE) ssr.run_resolve(graph, assr_params, draw_params, gbuffer);
    if (mask & VKRH_STAGE_GTAO_MAIN_ONLY)
      gtao.add_main_pass(graph, gtao_params, gbuffer.depth, gbuffer.normal, gbuffer.material, ssr.get_preintegrated_pdf());
    if (mask & VKRH_STAGE_GTAO) {                                                           // main.cpp:384-388
      gtao.add_main_pass(graph, gtao_params, gbuffer.depth, gbuffer.normal, gbuffer.material, ssr.get_preintegrated_pdf());
      gtao.add_filter_pass(graph, gtao_params, gbuffer.depth);
      gtao.add_accumulate_pass(graph, draw_params, gbuffer);
    }
    // ---- passes the reference ships but never records ----
    if (mask & VKRH_STAGE_GTAO_GRAPHICS) {
      gtao.add_main_pass_graphics(graph, gtao_params, gbuffer.depth, gbuffer.normal);
      gtao.add_filter_pass(graph, gtao_params, gbuffer.depth);
      const GTAOReprojection reprojection{prev_view * glm::inverse(view), fazz.x, fazz.y, fazz.z, fazz.w};
      gtao.add_reprojection_pass(graph, reprojection, gbuffer.depth, gbuffer.prev_depth);
    }
    if (mask & VKRH_STAGE_GTAO_DEINTERLEAVED) {
      gtao.deinterleave_depth(graph, gbuffer.depth);
      gtao.add_main_pass_deinterleaved(graph, gtao_params, gbuffer.normal);
    }
    if (mask & VKRH_STAGE_SCREEN_TRACE) {
      const ScreenTraceParams st_params{normal_mat, fazz.x, fazz.y, fazz.z, fazz.w};
      screen_trace.add_main_pass(graph, st_params, gbuffer.depth, gbuffer.normal, gbuffer.albedo, gbuffer.material);
      screen_trace.add_filter_pass(graph, st_params, gbuffer.depth);
      screen_trace.add_accumulate_pass(graph, st_params, gbuffer.depth, gbuffer.prev_depth);
    }
    // main.cpp:343,390-391: shading composes albedo / AO / reflections into color_out_tex, which TAA
    // resolves.  Without the shading stage TAA resolves the albedo attachment (the headline
    // composite of BASELINE.json is the nine passes without shading).
    if (mask & VKRH_STAGE_SHADING) {
      shading_pass.update_params(view, glm::mat4{1.f}, fazz.x, fazz.y, fazz.z, fazz.w);
      shading_pass.draw(graph, gbuffer, shadow_map, gtao.accumulated_ao, ssr.get_preintegrated_brdf(), ssr.get_blurred(), color_out_tex);
    }
    if (mask & VKRH_STAGE_TAA) taa_pass.run(graph, gbuffer, (mask & VKRH_STAGE_SHADING) ? color_out_tex : gbuffer.albedo, draw_params);
    graph.submit();
    if ((mask & (VKRH_STAGE_GBUFFER | VKRH_STAGE_RASTER)) && gbuffer.pipelined) copy_depth_to_next_set();
    task_names.clear();
    for (const auto& n : graph.last_submitted_tasks()) { task_names += n; task_names += '\n'; }
    task_lanes.clear();
    for (uint32_t l : graph.last_submitted_lanes()) { task_lanes += std::to_string(l); task_lanes += ' '; }
  }

  // pipelined: the G-buffer's depth (mip 0 of the depth image) also into the second set — the G-buffer of the benchmark is one
  // static frame; a renderer with two frames in flight draws every frame's depth into the set that frame will use
  void copy_depth_to_next_set() {
    const vkr_img a = graph.get_image(gbuffer.depth)->describe(0, 1), b = graph.get_image(gbuffer.depth_next)->describe(0, 1);
    if (hipMemcpy2DAsync(b.base, b.pitch_bytes[0], a.base, a.pitch_bytes[0], size_t(a.width) * 4, a.height, hipMemcpyDeviceToDevice, (hipStream_t)cfg.stream) != hipSuccess)
      throw std::runtime_error{"pipelined frame: copy of the depth into the second set failed"};
  }

  // TraceParams as run() hands them to the SSR passes (main.cpp:368-373), for the tiled frame's deferred hit-normal test
  vkr_trace_params trace_params() const {
    vkr_trace_params t {};
    const glm::mat4 normal_mat = glm::transpose(glm::inverse(view));
    std::memcpy(t.normal_mat.m, &normal_mat, sizeof(t.normal_mat.m));
    const glm::vec4 fazz = draw_params.fovy_aspect_znear_zfar;
    t.fovy = fazz.x; t.aspect = fazz.y; t.znear = fazz.z; t.zfar = fazz.w;
    return t;
  }

  void end_frame(bool swap_depth) {  // main.cpp:416-420
    if (swap_depth) graph.remap(gbuffer.depth, gbuffer.prev_depth);
    graph.remap(gtao.output, gtao.prev_frame);  // main.cpp:417: the reprojection variant's history (gtao.cpp:241-284)
    taa_pass.remap_targets(graph);
    ssr.remap_images(graph);
    gtao.remap(graph);
  }

  uint32_t hiz_gathered_mips = 4;  // tiled: view mips 0..3 of frame_hiz (image mips 1..4) arrive by all-gather

  rendergraph::ImageResourceId lookup(const std::string& name) {
    static const std::map<std::string, int> ids = {
        {"depth", 0}, {"prev_depth", 1}, {"normal", 2}, {"albedo", 3}, {"material", 4}, {"velocity", 5}, {"dn", 6}, {"dv", 7},
        {"raw", 8}, {"filtered", 9}, {"acc_ao", 10}, {"acc_hist", 11}, {"rays", 12}, {"reflections", 13}, {"blurred", 14},
        {"blurred_hist", 15}, {"pdf", 16}, {"taa_hist", 17}, {"taa_target", 18}, {"frame_hiz", 19}, {"frame_normals", 20},
        {"frame_albedo", 21}, {"color_out", 22}, {"brdf", 23}, {"ao_prev_frame", 24}, {"ao_output", 25}, {"deinterleaved_depth", 26},
        {"st_raw", 27}, {"st_filtered", 28}, {"st_accumulated", 29}, {"pend_mask", 30}};
    auto it = ids.find(name);
    if (it == ids.end()) throw std::runtime_error{"vkrh_image: unknown image '" + name + "'"};
    switch (it->second) {
      case 0: return gbuffer.depth; case 1: return gbuffer.prev_depth; case 2: return gbuffer.normal; case 3: return gbuffer.albedo;
      case 4: return gbuffer.material; case 5: return gbuffer.velocity_vectors; case 6: return gbuffer.downsampled_normals;
      case 7: return gbuffer.downsampled_velocity_vectors; case 8: return gtao.raw; case 9: return gtao.filtered;
      case 10: return gtao.accumulated_ao; case 11: return gtao.accumulated_history; case 12: return ssr.get_rays();
      case 13: return ssr.get_ouput(); case 14: return ssr.get_blurred(); case 15: return ssr.get_blurred_history();
      case 16: return ssr.get_preintegrated_pdf(); case 17: return taa_pass.get_history(); case 18: return taa_pass.get_output();
      case 19: return gbuffer.frame_hiz; case 20: return gbuffer.frame_normals; case 21: return gbuffer.frame_albedo;
      case 22: return color_out_tex; case 23: return ssr.get_preintegrated_brdf(); case 24: return gtao.prev_frame;
      case 25: return gtao.output; case 26: return gtao.deinterleaved_depth; case 27: return screen_trace.raw;
      case 28: return screen_trace.filtered;
      case 30: if (!gbuffer.normals_by_request) throw std::runtime_error{"vkrh_image: 'pend_mask' only exists with hit normals by request"}; return gbuffer.pend_mask;
      default: return screen_trace.accumulated;
    }
  }
};


// ---- the tiled frame: strips + RCCL (frame.hpp) ----------------------------------------------------------------------
struct TiledFrame {
  vkrh_tiled_config cfg;
  uint32_t W, H, th, y0, wy0, wh;  // tile rows [y0, y0 + th), window rows [wy0, wy0 + wh)
  bool tiled;
  std::unique_ptr<PostFxFrame> frame;
  hipStream_t compute = nullptr, xchg = nullptr;
  hipEvent_t ev_ready[5] {}, ev_done[5] {};  // 0 hiz, 1 albedo, 2 taa, 3 ao, 4 ssr
  bool halo_in_flight[3] {false, false, false};
  struct HaloBuf { void* send = nullptr; void* recv = nullptr; uint64_t bytes = 0; };
  HaloBuf halo[3][2];  // [surface][0: neighbour above, 1: neighbour below]

  static void check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error {std::string {what} + ": " + hipGetErrorString(e)}; }

  std::vector<uint32_t> bounds;   // world + 1 strip boundaries (rows)
  bool uniform = true;             // equal strips: ncclAllGather; otherwise vkr_all_gather_v
  std::vector<std::vector<uint64_t>> gather_offsets[2];  // [which][part][world + 1]: byte offsets of the ranks' shares

  explicit TiledFrame(const vkrh_tiled_config& c) : cfg {c} {
    W = c.full_width; H = c.full_height;
    if (c.world == 0 || c.rank >= c.world) throw std::runtime_error {"vkrh_tiled_create: rank outside the world"};
    tiled = c.world > 1 || c.force_tiled;
    const uint32_t k = c.gathered_mips;
    bounds.resize(c.world + 1);
    if (c.row_bounds) {
      for (uint32_t r = 0; r <= c.world; r++) bounds[r] = c.row_bounds[r];
      if (bounds[0] != 0 || bounds[c.world] != H) throw std::runtime_error {"vkrh_tiled_create: row_bounds must run from 0 to the frame height"};
      for (uint32_t r = 0; r < c.world; r++) {
        if (bounds[r + 1] <= bounds[r]) throw std::runtime_error {"vkrh_tiled_create: row_bounds must increase"};
        if (bounds[r + 1] - bounds[r] != H / c.world || H % c.world) uniform = false;
      }
    } else {
      if (H % c.world) throw std::runtime_error {"vkrh_tiled_create: the frame height must divide by the number of ranks"};
      for (uint32_t r = 0; r <= c.world; r++) bounds[r] = r * (H / c.world);
    }
    if (getenv("VKR_TILED_FORCE_GATHER_V")) uniform = false;  // tests: the broadcast-based gather also for equal strips
    cfg.row_bounds = nullptr;  // the caller's array need not outlive the call
    th = bounds[c.rank + 1] - bounds[c.rank]; y0 = bounds[c.rank];
    if (tiled && (k < 1 || k > 4 || W % (1u << k) || (c.world > 1 && c.halo % (1u << k))))
      throw std::runtime_error {"vkrh_tiled_create: tile extent and halo must be multiples of 2^gathered_mips (1..4)"};
    for (uint32_t r = 0; tiled && r <= c.world; r++)
      if (bounds[r] % (1u << k) || (bounds[r] & 1u)) throw std::runtime_error {"vkrh_tiled_create: strip bounds must be multiples of 2^gathered_mips (1..4)"};
    for (uint32_t r = 0; c.world > 1 && r < c.world; r++)
      if (c.halo > bounds[r + 1] - bounds[r] || (c.halo & 1u)) throw std::runtime_error {"vkrh_tiled_create: halo must be even and no larger than a strip"};
    const uint32_t halo_px = c.world > 1 ? c.halo : 0;
    wy0 = y0 >= halo_px ? y0 - halo_px : 0;
    const uint32_t wy1 = std::min(H, y0 + th + halo_px);
    wh = wy1 - wy0;
    compute = (hipStream_t)c.stream;
    vkrh_config fc {W, H, 0, (int32_t)wy0, W, wh, tiled ? 1u : 0u, c.stream};
    frame.reset(new PostFxFrame(fc));
    frame->hiz_gathered_mips = tiled ? k : 4;
    if (normals_by_request()) frame->gbuffer.enable_normal_requests(frame->graph, wy0 / 2, (wy0 + wh) / 2);
    if (tiled && c.world > 1 && !env_set("VKR_TILED_WHOLE_WINDOW")) clip_outputs();
    // Two frames in flight (VKR_TILED_PIPELINE=1, native wire, hit colours by request): the next frame's downsample and depth
    // all-gather start right after this frame's trace and travel while its GTAO, TAA, resolve and blur run (pipelined_step).
    pipeline_enabled = tiled && c.world > 1 && by_request() && env_set("VKR_TILED_PIPELINE");
    if (pipeline_enabled) frame->gbuffer.enable_pipelining(frame->graph);
    if (tiled) {
      // the exchanges' kernels (a few workgroups each) must not queue behind a frame's worth of compute waves
      int prio_low = 0, prio_high = 0;
      check(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high), "stream priority range");
      check(hipStreamCreateWithPriority(&xchg, hipStreamNonBlocking, prio_high), "exchange stream");
      for (auto& e : ev_ready) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
      for (auto& e : ev_done) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
      if (by_request()) hit_init();
      for (int s = 0; s < 3; s++)
        for (int n = 0; n < 2; n++) {
          if (!neighbour(n, nullptr)) continue;
          const vkr_img d = surface(s);
          halo[s][n].bytes = uint64_t(halo_px >> halo_dv(s)) * d.width * vkr_format_bytes(d.format);
          halo[s][n].send = gpu::device_alloc(halo[s][n].bytes);
          halo[s][n].recv = gpu::device_alloc(halo[s][n].bytes);
        }
    }
  }
  bool pipeline_enabled = false, primed = false;
  static bool env_set(const char* name) { const char* v = getenv(name); return v && *v && *v != '0'; }
  // Which rows of its window a rank has to COMPUTE (gpu::Image::set_store_rows; VKR_TILED_WHOLE_WINDOW=1 keeps round 3's "every
  // pass on the whole window").  The three history surfaces and the filtered AO: the strip itself — the halo rows of a history
  // arrive from the neighbours before the next frame reads them (copy_halo overwrote what the pass had computed there anyway),
  // and the accumulation reads the filtered AO at its own pixel only.  The resolved reflections: the strip and the blur's apron
  // (blur.comp: taps up to 11 texels away; ssr.hip BLUR_R).  The rays — with their pending mask / data and the (occlusion, pdf)
  // image the trace shares with GTAO main — two rows more: the resolve reads its four neighbours (filter.comp:112-134), GTAO's
  // filter taps reach two rows (filter.comp:17-51); the request / reply round and the deferred normal test walk the same rows
  // (rays_img() etc. describe the store rows).  The downsample stays whole-window: everything above reads its halo.
  void clip_outputs() {
    auto& g = frame->graph;
    const uint32_t top = y0 - wy0;  // rows of halo above the strip inside the window
    const auto rows = [&](rendergraph::ImageResourceId id, uint32_t shift, uint32_t apron) {
      gpu::Image& img = *g.get_image(id);
      const uint32_t first = top >> shift, last = (top + th) >> shift, h = img.get_info().height;
      const uint32_t lo = first > apron ? first - apron : 0, hi = std::min(h, last + apron);
      img.set_store_rows(lo, hi - lo);
    };
    rows(frame->taa_pass.get_output(), 0, 0); rows(frame->taa_pass.get_history(), 0, 0);
    rows(frame->gtao.accumulated_ao, 1, 0); rows(frame->gtao.accumulated_history, 1, 0); rows(frame->gtao.filtered, 1, 0);
    rows(frame->ssr.get_blurred(), 1, 0); rows(frame->ssr.get_blurred_history(), 1, 0);
    rows(frame->ssr.get_ouput(), 1, 12);
    rows(frame->ssr.get_rays(), 1, 14); rows(frame->gtao.raw, 1, 14);
    if (normals_by_request()) { rows(frame->gbuffer.pend_mask, 1, 14); rows(frame->gbuffer.pend_data, 1, 14); }
  }
  void drop_wait_marks() {
    for (auto& v : wait_marks) { for (auto& m : v) { (void)hipEventDestroy(m.first); (void)hipEventDestroy(m.second); } v.clear(); }
  }
  ~TiledFrame() {
    if (compute) (void)hipStreamSynchronize(compute);
    drop_wait_marks();
    if (xchg) { (void)hipStreamSynchronize(xchg); (void)hipStreamDestroy(xchg); }
    for (auto e : ev_ready) if (e) (void)hipEventDestroy(e);
    for (auto e : ev_done) if (e) (void)hipEventDestroy(e);
    for (auto& s : halo) for (auto& b : s) { gpu::device_free(b.send); gpu::device_free(b.recv); }
    hit_release();
  }

  // ---- geometry -------------------------------------------------------------------------------------------------
  bool neighbour(int n, int* rank) const {  // 0: the strip above, 1: the strip below
    if (cfg.world < 2) return false;
    if (n == 0 ? cfg.rank == 0 : cfg.rank + 1 == cfg.world) return false;
    if (rank) *rank = n == 0 ? int(cfg.rank) - 1 : int(cfg.rank) + 1;
    return true;
  }
  static uint32_t halo_dv(int s) { return s == VKRH_HALO_TAA ? 0u : 1u; }
  // the pass OUTPUT of each history surface (it becomes the history at the end-of-frame remap)
  vkr_img surface(int s) {
    const auto id = s == VKRH_HALO_TAA ? frame->taa_pass.get_output() : s == VKRH_HALO_AO ? frame->gtao.accumulated_ao : frame->ssr.get_blurred();
    return frame->graph.get_image(id)->describe(0, 1);
  }
  // pack (to_buffers) / unpack the halo rows of surface s: one vkr_copy_rects launch on the compute stream.  A refresh is
  // packed right after the pass that wrote the surface and unpacked in the NEXT frame (or in flush()), i.e. after
  // end_frame() has swapped output and history: by then the output id names the old history image, which the pass is
  // about to overwrite.  The rows belong into the very image they were packed from — now the history the pass reads —
  // so the descriptor is taken at pack time and kept for the unpack.
  vkr_img packed_from[3] {};
  void copy_halo(int s, bool to_buffers) {
    gpu::TraceRange range {to_buffers ? "halo pack" : "halo unpack"};
    if (to_buffers) packed_from[s] = surface(s);
    const vkr_img d = packed_from[s];
    const uint32_t dv = halo_dv(s), bpp = vkr_format_bytes(d.format), rows = cfg.halo >> dv, row_bytes = d.width * bpp;
    const uint32_t ty0 = (y0 >> dv) - uint32_t(d.origin_y), tth = th >> dv;  // the tile's first row inside the window image
    vkr_rect_copy rc[2];
    uint32_t n = 0;
    for (int nb = 0; nb < 2; nb++) {
      if (!neighbour(nb, nullptr)) continue;
      // send: my first / last `rows` tile rows (they lie in that neighbour's halo); receive: the rows just outside my tile
      const uint32_t send_row = nb == 0 ? ty0 : ty0 + tth - rows, recv_row = nb == 0 ? ty0 - rows : ty0 + tth;
      const uint64_t img = (uint64_t)(uintptr_t)d.base;
      if (to_buffers) rc[n++] = vkr_rect_copy {img + uint64_t(send_row) * d.pitch_bytes[0], (uint64_t)(uintptr_t)halo[s][nb].send, d.pitch_bytes[0], row_bytes, row_bytes, rows};
      else rc[n++] = vkr_rect_copy {(uint64_t)(uintptr_t)halo[s][nb].recv, img + uint64_t(recv_row) * d.pitch_bytes[0], row_bytes, d.pitch_bytes[0], row_bytes, rows};
    }
    if (n && vkr_copy_rects(rc, n, compute) != 0) throw std::runtime_error {std::string {"copy_rects: "} + vkr_last_error()};
  }
  uint32_t halo_peers(int s, vkr_halo_peer* out) {
    uint32_t n = 0;
    for (int nb = 0; nb < 2; nb++) {
      int peer;
      if (!neighbour(nb, &peer)) continue;
      out[n++] = vkr_halo_peer {peer, 0u, halo[s][nb].send, halo[s][nb].bytes, halo[s][nb].recv, halo[s][nb].bytes};
    }
    return n;
  }
  // A strip's rows of a whole-frame surface are contiguous in the window image AND in the frame image (same width, same
  // pitch), so every surface is gathered in place: send = the tile's rows where they lie, recv = the frame image.
  uint32_t gather_parts(int which, vkr_gather_part* out, bool next_set = false) {
    auto& offs = gather_offsets[which];
    offs.clear();
    auto part = [&](rendergraph::ImageResourceId src, uint32_t src_mip, rendergraph::ImageResourceId dst, uint32_t dst_mip, uint32_t dv) {
      const vkr_img s = frame->graph.get_image(src)->describe(src_mip, 1), d = frame->graph.get_image(dst)->describe(dst_mip, 1);
      const uint32_t rows = th >> dv;
      if (s.pitch_bytes[0] != d.pitch_bytes[0] || d.height != (H >> dv) || s.width != d.width || d.origin_y != 0)
        throw std::runtime_error {"tiled frame: window and whole-frame images must share width and row pitch"};
      const uint32_t ly = (y0 >> dv) - uint32_t(s.origin_y);
      std::vector<uint64_t> o(cfg.world + 1);
      for (uint32_t r = 0; r <= cfg.world; r++) o[r] = uint64_t(bounds[r] >> dv) * s.pitch_bytes[0];
      offs.push_back(std::move(o));
      return vkr_gather_part {(const uint8_t*)s.base + uint64_t(ly) * s.pitch_bytes[0], d.base, uint64_t(rows) * s.pitch_bytes[0]};
    };
    uint32_t n = 0;
    auto& g = frame->gbuffer;
    if (which == VKRH_GATHER_HIZ) {  // (next_set: what the pipelined frame has just downsampled for the frame after this one)
      for (uint32_t m = 1; m <= cfg.gathered_mips; m++) out[n++] = part(next_set ? g.depth_next : g.depth, m, g.frame_hiz, m - 1, m);
      if (!normals_by_request()) out[n++] = part(next_set ? g.downsampled_normals_next : g.downsampled_normals, 0, g.frame_normals, 0, 1);
    } else {
      out[n++] = part(g.albedo, 0, g.frame_albedo, 0, 0);
    }
    return n;
  }

  // ---- ordering between the compute and the exchange stream ----------------------------------------------------------
  void start(int slot, const std::function<int()>& issue) {
    if (!cfg.comm) return;  // harness mode: the caller moves the bytes between phases
    check(hipEventRecord(ev_ready[slot], compute), "event record");
    check(hipStreamWaitEvent(xchg, ev_ready[slot], 0), "stream wait");
    if (issue() != 0) throw std::runtime_error {std::string {"exchange: "} + vkr_last_error()};
    check(hipEventRecord(ev_done[slot], xchg), "event record");
  }
  // Exposed waiting (diagnostics, vkrh_tiled_wait_times): with wait timing on, an event pair on the compute stream brackets
  // every wait for an exchange — the second event cannot complete before the exchange has, so the pair measures how long
  // the compute stream actually stood still for it (0 when the bytes had already arrived).
  bool time_waits = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> wait_marks[5];
  void wait(int slot) {
    if (!cfg.comm) return;
    hipEvent_t before = nullptr, after = nullptr;
    if (time_waits) {
      check(hipEventCreate(&before), "event"); check(hipEventCreate(&after), "event");
      check(hipEventRecord(before, compute), "event record");
    }
    check(hipStreamWaitEvent(compute, ev_done[slot], 0), "stream wait");
    if (time_waits) {
      check(hipEventRecord(after, compute), "event record");
      wait_marks[slot].emplace_back(before, after);
    }
  }
  // total exposed wait per exchange since the last call, ms: [hiz, albedo, taa halo, ao halo, ssr halo]; synchronises
  void collect_waits(float out[5]) {
    check(hipStreamSynchronize(compute), "synchronize");
    for (int s = 0; s < 5; s++) {
      out[s] = 0.0f;
      for (auto& m : wait_marks[s]) {
        float ms = 0.0f;
        check(hipEventElapsedTime(&ms, m.first, m.second), "event elapsed");
        out[s] += ms;
        (void)hipEventDestroy(m.first); (void)hipEventDestroy(m.second);
      }
      wait_marks[s].clear();
    }
  }
  void start_gather(int which, bool next_set = false) {
    gpu::TraceRange range {which == VKRH_GATHER_HIZ ? "all-gather Hi-Z" : "all-gather albedo"};
    start(which, [&] {
      vkr_gather_part p[8];
      const uint32_t n = gather_parts(which, p, next_set);
      if (uniform) return vkr_all_gather(cfg.comm, p, n, xchg);
      vkr_gather_v_part v[8];  // strips of different heights: every share at its own offset of the frame image
      for (uint32_t i = 0; i < n; i++) v[i] = vkr_gather_v_part {p[i].send, p[i].recv, gather_offsets[which][i].data()};
      return vkr_all_gather_v(cfg.comm, v, n, xchg);
    });
  }
  void start_halo(int s) {
    gpu::TraceRange range {"halo exchange"};
    halo_in_flight[s] = true;
    start(2 + s, [&] { vkr_halo_peer p[2]; const uint32_t n = halo_peers(s, p); return vkr_halo_exchange(cfg.comm, p, n, xchg); });
  }
  void finish_halo(int s) {  // the refresh of surface s started in the previous frame: wait for it, then scatter it into the ring
    if (!halo_in_flight[s]) return;
    wait(2 + s);
    copy_halo(s, false);
    halo_in_flight[s] = false;
  }

  // ---- hit colours by request / reply (frame.hpp; csrc/hit_exchange.hip) ------------------------------------------------
  struct HitState {
    // device, every part 64-byte aligned (RCCL moves the counts): counts @0, reply errors @HIT_ERRORS,
    // the gathered world x world matrix @HIT_MATRIX
    uint32_t* counts = nullptr;
    uint32_t* workspace = nullptr;  // VKR_HIT_WORKSPACE_WORDS: what pass 1 of vkr_hit_requests leaves for pass 2
    uint32_t* host_counts = nullptr;  // pinned: world * world + 1
    vkr_hit_request* req_out = nullptr; uint8_t* reply_in = nullptr; uint64_t cap_out = 0;   // what I ask / get back
    vkr_hit_request* req_in = nullptr; uint8_t* reply_out = nullptr; uint64_t cap_in = 0;    // what I am asked / answer
    std::vector<uint32_t> out_seg, in_seg;  // [world + 1]: my requests for owner o / the requests of rank r for me, as runs
    uint64_t wire_bytes = 0;
    bool counted = false;
    // Native wire without a host round trip: the room of every rank-to-owner segment THIS frame, derived on every rank from
    // LAST frame's world x world counts (identical everywhere: the counts are all-gathered).  With it the whole round — requests,
    // both exchanges, replies, scatter — is enqueued behind the trace at once; the host only looks at this frame's counts
    // afterwards (phase 4) and repeats the round exactly if a segment was too small.  Empty: no previous frame yet.
    std::vector<uint32_t> cap_matrix;
    std::vector<uint32_t> caps_out;     // [world]: row `rank` of cap_matrix (what vkr_hit_requests_bounded takes)
    hipEvent_t ev_counts = nullptr;     // this frame's counts (and last frame's reply errors) have reached host_counts
    bool speculative = false;           // this frame's round went out on capacities
    uint32_t cap_percent = 125;         // room = this share of last frame's count (+ 64, rounded up to 64); VKR_HIT_CAP_PERCENT
    uint64_t rounds_speculative = 0, rounds_exact = 0, rounds_repeated = 0;
  } hit;
  // The trace in two stages around the gather (needs the pending-ray images of the request / reply mode): VKR_TILED_LOCAL_FIRST=1.
  // OFF by default: measured on a 15360 x 1080 strip of config 4 (tools/trace_local_probe.py) the head parks 70 - 76 % of the
  // rays — a march climbs past the gathered levels after four skipped tiles — so the resume launch repeats most of the work
  // through 80-byte records: head 0.42 + resume 0.67 ms against 0.70 ms for the one launch, more than the 0.16 - 0.28 ms of
  // exposed gather it can hide.
  bool local_first() const { return normals_by_request() && local_first_enabled; }
  bool local_first_enabled = getenv("VKR_TILED_LOCAL_FIRST") != nullptr && std::string {getenv("VKR_TILED_LOCAL_FIRST")} == "1";
  bool by_request() const { return tiled && cfg.world > 1 && cfg.albedo_by_gather != 1; }          // hit colours
  bool normals_by_request() const { return tiled && cfg.world > 1 && cfg.albedo_by_gather == 0; }   // ... and hit normals
  vkr_img dn_img() { return frame->graph.get_image(frame->gbuffer.downsampled_normals)->describe(0, 1); }
  vkr_img frame_normals_img() { return frame->graph.get_image(frame->gbuffer.frame_normals)->describe(0, 1); }
  vkr_img pend_mask_img() { return frame->graph.get_image(frame->gbuffer.pend_mask)->describe_store(0, 1); }
  vkr_img pend_data_img() { return frame->graph.get_image(frame->gbuffer.pend_data)->describe_store(0, 1); }
  // what vkr_hit_requests walks: the rays for the albedo rows, the pending rays of the windowed trace for the normal rows
  struct HitSources { vkr_img rays, mask, data; vkr_hit_sources src; };
  void hit_sources(HitSources& h) {
    h.rays = rays_img();
    h.src = vkr_hit_sources {&h.rays, W, H, wy0, wy0 + wh, nullptr, nullptr, 0, 0, 0, 0};
    if (normals_by_request()) {
      h.mask = pend_mask_img(); h.data = pend_data_img();
      h.src.pending_mask = &h.mask; h.src.pending_data = &h.data;
      h.src.normal_width = W / 2; h.src.normal_height = H / 2; h.src.normal_row0 = wy0 / 2; h.src.normal_row1 = (wy0 + wh) / 2;
    }
  }
  vkr_img rays_img() { return frame->graph.get_image(frame->ssr.get_rays())->describe_store(0, 1); }
  vkr_img albedo_img() { return frame->graph.get_image(frame->gbuffer.albedo)->describe(0, 1); }
  vkr_img frame_albedo_img() { return frame->graph.get_image(frame->gbuffer.frame_albedo)->describe(0, 1); }
  void hit_init() {
    const uint32_t w = cfg.world;
    if (w > 16) throw std::runtime_error {"tiled frame: the request / reply exchange is laid out for at most 16 ranks"};
    hit.counts = (uint32_t*)gpu::device_alloc(sizeof(uint32_t) * HIT_WORDS);
    check(hipMemset(hit.counts, 0, sizeof(uint32_t) * HIT_WORDS), "memset");  // (the error word is read before the first count clears it)
    hit.workspace = (uint32_t*)gpu::device_alloc(sizeof(uint32_t) * VKR_HIT_WORKSPACE_WORDS);
    check(hipHostMalloc((void**)&hit.host_counts, sizeof(uint32_t) * (w * w + 4), hipHostMallocDefault), "pinned counts");
    hit.out_seg.assign(w + 1, 0); hit.in_seg.assign(w + 1, 0);
    // room for what this rank can ask for — every ray of its window ending on another strip — and the same for what it
    // may be asked (anything beyond grows, see grow())
    // (the worst case — every ray of the window ending on another strip, (4 + 16) B x 2 directions x 2 surfaces — was
    // 370 MB per rank at c4 for an exchange that moves 10 - 30 MB; a quarter of the rays is still 4x what the frames ask, and grow() covers the rest)
    const uint64_t worst = ((normals_by_request() ? 2ull : 1ull) * (W / 2) * (wh / 2)) / 4 + 4096;
    uint64_t cap = 0;
    grow((void**)&hit.req_out, &cap, worst, sizeof(vkr_hit_request));
    grow((void**)&hit.reply_in, &hit.cap_out, worst, VKR_HIT_REPLY_BYTES);
    cap = 0;
    grow((void**)&hit.req_in, &cap, worst, sizeof(vkr_hit_request));
    grow((void**)&hit.reply_out, &hit.cap_in, worst, VKR_HIT_REPLY_BYTES);
    grew = false;  // before the first frame: the caller synchronises after set-up (prepare), nothing is in flight
    check(hipEventCreateWithFlags(&hit.ev_counts, hipEventDisableTiming), "event");
    if (const char* e = getenv("VKR_HIT_CAP_PERCENT")) hit.cap_percent = (uint32_t)std::max(1, atoi(e));
  }
  void hit_release() {
    gpu::device_free(hit.counts);
    gpu::device_free(hit.workspace);
    if (hit.host_counts) (void)hipHostFree(hit.host_counts);
    if (hit.ev_counts) (void)hipEventDestroy(hit.ev_counts);
    gpu::device_free(hit.req_out); gpu::device_free(hit.reply_in); gpu::device_free(hit.req_in); gpu::device_free(hit.reply_out);
  }
  // own rows of the whole-frame albedo: the window's rows are copied where the all-gather would have put the tile's
  void hit_local_rows(hipStream_t s) {
    const vkr_img a = albedo_img(), f = frame_albedo_img();
    if (a.pitch_bytes[0] != f.pitch_bytes[0] || a.width != f.width) throw std::runtime_error {"tiled frame: window and whole-frame albedo must share width and row pitch"};
    check(hipMemcpyAsync((uint8_t*)f.base + uint64_t(a.origin_y) * f.pitch_bytes[0], a.base, uint64_t(a.height) * a.pitch_bytes[0], hipMemcpyDeviceToDevice, s), "albedo rows");
    if (normals_by_request()) {
      const vkr_img n = dn_img(), fn = frame_normals_img();
      if (n.pitch_bytes[0] != fn.pitch_bytes[0] || n.width != fn.width) throw std::runtime_error {"tiled frame: window and whole-frame normals must share width and row pitch"};
      check(hipMemcpyAsync((uint8_t*)fn.base + uint64_t(n.origin_y) * fn.pitch_bytes[0], n.base, uint64_t(n.height) * n.pitch_bytes[0], hipMemcpyDeviceToDevice, s), "normal rows");
    }
  }
  // pass 1, behind the trace on stream s (the harness: the compute stream; with a communicator: the exchange stream, so that
  // GTAO starts right after the trace and the count runs beside it)
  void hit_count(hipStream_t s) {
    const uint32_t w = cfg.world;
    check(hipMemsetAsync(hit.counts, 0, sizeof(uint32_t) * HIT_MATRIX, s), "memset");
    HitSources h;
    hit_sources(h);
    if (vkr_hit_requests(&h.src, bounds.data(), w, hit.counts, hit.workspace, nullptr, nullptr, s) != 0)
      throw std::runtime_error {std::string {"hit_requests: "} + vkr_last_error()};
    hit.counted = true;
  }
  // An allocator may initialise what it hands out asynchronously on the COMPUTE stream (the torch-backed one zero-fills
  // there).  Buffers the exchange stream writes are therefore allocated before the first frame, for the worst case of
  // this rank's own requests; should a rank ever be asked for more than that, the exchange stream is ordered behind the
  // compute stream once, so that a late fill cannot wipe what the exchange has already landed (found by the 4-process
  // wire test under load: replies zeroed in the frame that allocated them).
  bool grew = false;
  void grow(void** p, uint64_t* cap, uint64_t need, uint64_t elem) {
    if (need <= *cap) return;
    // a round that went out on capacities may still be in flight on the exchange stream (the host no longer waits for it): what
    // it reads and writes must not go back to the allocator under it.  Rare: a buffer grows by a quarter more than it needs.
    if (*p && xchg) check(hipStreamSynchronize(xchg), "synchronize");
    gpu::device_free(*p);
    *cap = need + need / 4 + 1024;
    *p = gpu::device_alloc(*cap * elem);
    grew = true;
  }
  void order_exchange_behind_allocations(hipStream_t s) {
    if (!grew || s == compute) { grew = false; return; }
    grew = false;
    check(hipEventRecord(ev_ready[VKRH_GATHER_ALBEDO], compute), "event record");
    check(hipStreamWaitEvent(s, ev_ready[VKRH_GATHER_ALBEDO], 0), "stream wait");
  }
  // pass 2 on stream s, given everybody's counts — or, bounded, everybody's segment capacities: fills req_out and returns the
  // peer list of the request exchange
  uint32_t hit_write(const uint32_t* matrix, hipStream_t s, vkr_halo_peer* peers, bool bounded = false) {
    const uint32_t w = cfg.world, me = cfg.rank;
    for (uint32_t o = 0; o < w; o++) hit.out_seg[o + 1] = hit.out_seg[o] + matrix[me * w + o];
    for (uint32_t r = 0; r < w; r++) hit.in_seg[r + 1] = hit.in_seg[r] + matrix[r * w + me];
    if (matrix[me * w + me]) throw std::runtime_error {"tiled frame: a rank requested hit colours from itself"};
    uint64_t cap = hit.cap_out;
    grow((void**)&hit.req_out, &cap, hit.out_seg[w], sizeof(vkr_hit_request));
    grow((void**)&hit.reply_in, &hit.cap_out, hit.out_seg[w], VKR_HIT_REPLY_BYTES);
    cap = hit.cap_in;
    grow((void**)&hit.req_in, &cap, hit.in_seg[w], sizeof(vkr_hit_request));
    grow((void**)&hit.reply_out, &hit.cap_in, hit.in_seg[w], VKR_HIT_REPLY_BYTES);
    order_exchange_behind_allocations(s);
    if (hit.out_seg[w]) {
      HitSources h;
      hit_sources(h);
      int rc;
      // Every slot starts as "no request".  Bounded segments have unused room by construction; an exact round that REPEATS a
      // bounded one may write fewer requests than it counted, because the deferred normal test of the first round has
      // meanwhile turned some provisional hits into misses (their albedo texels are no longer asked for).
      check(hipMemsetAsync(hit.req_out, 0xFF, uint64_t(hit.out_seg[w]) * sizeof(vkr_hit_request), s), "memset");
      if (bounded) {
        hit.caps_out.assign(matrix + me * w, matrix + me * w + w);
        rc = vkr_hit_requests_bounded(&h.src, bounds.data(), w, hit.workspace, hit.out_seg.data(), hit.caps_out.data(), hit.req_out, hit.counts + HIT_DROPPED, s);
      } else {
        rc = vkr_hit_requests(&h.src, bounds.data(), w, hit.counts, hit.workspace, hit.out_seg.data(), hit.req_out, s);
      }
      if (rc != 0) throw std::runtime_error {std::string {"hit_requests: "} + vkr_last_error()};
    }
    uint32_t n = 0;
    hit.wire_bytes = 0;
    for (uint32_t p = 0; p < w; p++) {
      const uint64_t so = hit.out_seg[p + 1] - hit.out_seg[p], ri = hit.in_seg[p + 1] - hit.in_seg[p];
      if (p == me || (so == 0 && ri == 0)) continue;
      peers[n++] = vkr_halo_peer {int32_t(p), 0u, hit.req_out + hit.out_seg[p], so * sizeof(vkr_hit_request), hit.req_in + hit.in_seg[p], ri * sizeof(vkr_hit_request)};
      hit.wire_bytes += ri * sizeof(vkr_hit_request) + so * VKR_HIT_REPLY_BYTES;  // requests in now, replies to my requests later
    }
    return n;
  }
  // answers on stream s; returns the peer list of the way back
  uint32_t hit_reply(hipStream_t s, vkr_halo_peer* peers) {
    const uint32_t w = cfg.world, me = cfg.rank;
    const vkr_img a = albedo_img(), dn = dn_img();
    if (vkr_hit_reply(&a, normals_by_request() ? &dn : nullptr, hit.req_in, hit.in_seg[w], hit.reply_out, hit.counts + HIT_ERRORS, s) != 0)
      throw std::runtime_error {std::string {"hit_reply: "} + vkr_last_error()};
    uint32_t n = 0;
    for (uint32_t p = 0; p < w; p++) {
      const uint64_t so = hit.in_seg[p + 1] - hit.in_seg[p], ri = hit.out_seg[p + 1] - hit.out_seg[p];
      if (p == me || (so == 0 && ri == 0)) continue;
      peers[n++] = vkr_halo_peer {int32_t(p), 0u, hit.reply_out + uint64_t(hit.in_seg[p]) * VKR_HIT_REPLY_BYTES, so * VKR_HIT_REPLY_BYTES,
                                  hit.reply_in + uint64_t(hit.out_seg[p]) * VKR_HIT_REPLY_BYTES, ri * VKR_HIT_REPLY_BYTES};
    }
    return n;
  }
  void hit_scatter(hipStream_t s, bool bounded = false) {
    const vkr_img f = frame_albedo_img(), fn = frame_normals_img();
    if (vkr_hit_scatter(&f, normals_by_request() ? &fn : nullptr, hit.req_out, hit.reply_in, hit.out_seg[cfg.world], s) != 0)
      throw std::runtime_error {std::string {"hit_scatter: "} + vkr_last_error()};
    if (normals_by_request()) {  // the hit-normal test the windowed trace deferred: the footprints are complete now
      const vkr_img r = rays_img(), m = pend_mask_img(), d = pend_data_img();
      const vkr_trace_params tp = frame->trace_params();
      // (a bounded round that dropped requests of THIS rank has not brought every hit normal: the test waits for the exact round)
      if (vkr_sssr_validate_unless(&r, &m, &d, &fn, &tp, bounded ? hit.counts + HIT_DROPPED : nullptr, s) != 0) throw std::runtime_error {std::string {"sssr_validate: "} + vkr_last_error()};
    }
    hit.counted = false;
  }
  // The native exchange.  Behind the trace, on the exchange stream: count, all-gather of the counts, their copy to the host
  // (with the reply errors of the previous frame) and an event.  With capacities from the previous frame the whole round
  // follows at once (hit_round: requests into fixed-room segments, the two point-to-point exchanges at those sizes, replies,
  // scatter, the deferred normal test) and ends in ev_done[VKRH_GATHER_ALBEDO], which the filter waits for: no host round trip
  // between the trace and the filter.  hit_exchange_complete() is where the host looks at the counts: in the first frame it
  // is the round trip (called with GTAO queued, before anything else goes on the exchange stream: ADVICE r03 — the old
  // whole-stream synchronise also waited for GTAO and its halo exchange); later it only checks that no segment overflowed
  // and otherwise repeats the round exactly (every rank sees the same matrix and takes the same branch).
  void hit_exchange_native() {
    const uint32_t w = cfg.world;
    check(hipEventRecord(ev_ready[VKRH_GATHER_ALBEDO], compute), "event record");  // the trace (and its pending images) are complete
    check(hipStreamWaitEvent(xchg, ev_ready[VKRH_GATHER_ALBEDO], 0), "stream wait");
    // the error word of the previous frame's replies, before the memset of hit_count() clears it
    check(hipMemcpyAsync(hit.host_counts + w * w, hit.counts + HIT_ERRORS, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, xchg), "errors to host");
    hit_count(xchg);
    const vkr_gather_part part {hit.counts, hit.counts + HIT_MATRIX, uint64_t(w) * sizeof(uint32_t)};
    if (vkr_all_gather(cfg.comm, &part, 1, xchg) != 0) throw std::runtime_error {std::string {"exchange: "} + vkr_last_error()};
    check(hipMemcpyAsync(hit.host_counts, hit.counts + HIT_MATRIX, sizeof(uint32_t) * w * w, hipMemcpyDeviceToHost, xchg), "counts to host");
    check(hipEventRecord(hit.ev_counts, xchg), "event record");
    hit_pending = true;
    hit.speculative = !hit.cap_matrix.empty();
    if (hit.speculative) { hit_round(hit.cap_matrix.data(), true); hit.rounds_speculative++; }
  }
  // requests -> exchange -> replies -> exchange -> scatter (+ deferred normal test), all on the exchange stream
  void hit_round(const uint32_t* matrix, bool bounded) {
    vkr_halo_peer peers[HIT_PEERS];
    uint32_t n = hit_write(matrix, xchg, peers, bounded);
    if (vkr_halo_exchange(cfg.comm, peers, n, xchg) != 0) throw std::runtime_error {std::string {"exchange: "} + vkr_last_error()};
    n = hit_reply(xchg, peers);
    if (vkr_halo_exchange(cfg.comm, peers, n, xchg) != 0) throw std::runtime_error {std::string {"exchange: "} + vkr_last_error()};
    hit_scatter(xchg, bounded);
    check(hipEventRecord(ev_done[VKRH_GATHER_ALBEDO], xchg), "event record");
  }
  bool hit_pending = false;
  void hit_exchange_complete() {  // host-blocking on the counts only
    if (!hit_pending) return;
    gpu::TraceRange range {hit.speculative ? "hit colours: check the counts" : "hit colours: counts -> requests -> replies -> scatter"};
    hit_pending = false;
    const uint32_t w = cfg.world;
    check(hipEventSynchronize(hit.ev_counts), "event synchronize");
    if (hit.host_counts[w * w] != 0)
      throw std::runtime_error {"tiled frame: " + std::to_string(hit.host_counts[w * w]) + " hit-colour requests of the previous frame named texels their owner does not hold (the ranks' strips disagree); one of them: request " +
                                std::to_string(hit.host_counts[w * w + 1]) + " in slot " + std::to_string(hit.host_counts[w * w + 2]) + " of " + std::to_string(hit.in_seg[w])};
    bool overflow = false;
    for (uint32_t i = 0; hit.speculative && i < w * w; i++) overflow = overflow || hit.host_counts[i] > hit.cap_matrix[i];
    if (!hit.speculative || overflow) {
      hit_round(hit.host_counts, false);
      (overflow ? hit.rounds_repeated : hit.rounds_exact)++;
    }
    hit.cap_matrix.resize(w * w);
    hit_capacities(hit.host_counts, w, hit.cap_percent, hit.cap_matrix.data());
    hit.speculative = false;
  }
  // room for the next frame: last count + a quarter, in steps of 64; neighbours always keep a segment, a pair that asked for
  // nothing and is not adjacent keeps none (if it ever asks, that frame repeats its round)
  static void hit_capacities(const uint32_t* counts, uint32_t w, uint32_t percent, uint32_t* caps) {
    for (uint32_t r = 0; r < w; r++)
      for (uint32_t o = 0; o < w; o++) {
        const uint32_t c = counts[r * w + o];
        const bool adjacent = r + 1 == o || o + 1 == r;
        caps[r * w + o] = (r == o || (c == 0 && !adjacent)) ? 0u : uint32_t((uint64_t(c) * percent / 100 + 63) / 64 * 64 + 64);
      }
  }
  // Measurement (tools/wire_emulation.py): a frame that has so far been driven phase by phase by the in-process harness —
  // which left in its receive buffers exactly what real peers send, in segments laid out by hit_capacities(counts) — goes on
  // natively (vkrh_tiled_step) on `comm`, an emulated communicator (vkr_comm_create_emulated): the exchanges take their
  // time on the exchange stream and deliver what is already there.  `counts` is the world x world matrix of that last frame.
  void emulate_wire(void* comm, const uint32_t* counts) {
    if (!tiled || cfg.world < 2 || !comm) throw std::runtime_error {"vkrh_tiled_emulate_wire: needs a harness-driven frame of several ranks and a communicator"};
    check(hipStreamSynchronize(compute), "synchronize");
    if (xchg) check(hipStreamSynchronize(xchg), "synchronize");
    const bool again = cfg.comm != nullptr;  // a second call only swaps the communicator (another link rate): the frame's state stays
    cfg.comm = (vkr_comm*)comm;
    if (again) { flush(); check(hipStreamSynchronize(compute), "synchronize"); return; }
    if (by_request()) {
      if (!counts) throw std::runtime_error {"vkrh_tiled_emulate_wire: the counts of the last frame are needed (hit colours by request)"};
      const uint32_t w = cfg.world;
      check(hipMemcpy(hit.counts + HIT_MATRIX, counts, sizeof(uint32_t) * w * w, hipMemcpyHostToDevice), "counts");
      hit.cap_matrix.resize(w * w);
      hit_capacities(counts, w, hit.cap_percent, hit.cap_matrix.data());
      hit.speculative = false; hit_pending = false; hit.counted = false;
    }
  }
  static constexpr uint32_t HIT_PEERS = 16;
  static constexpr uint32_t HIT_ERRORS = 32, HIT_DROPPED = 40, HIT_MATRIX = 64, HIT_WORDS = 64 + 256;  // world <= 16; [0, HIT_MATRIX) is cleared by every count

  // ---- the frame, in phases (an exchange may only start / must be complete at a phase boundary) ------------------------
  void phase(uint32_t p) {
    if (!tiled) throw std::runtime_error {"vkrh_tiled_phase: this frame is not tiled (one rank without force_tiled has no phases: use vkrh_tiled_step)"};
    PostFxFrame& f = *frame;
    static const char* const names[VKRH_TILED_PHASES] = {"tiled: downsample | start gathers", "tiled: TAA | halo", "tiled: Hi-Z tail + trace",
                                                        "tiled: GTAO | halo | hit colours", "tiled: SSR filter + blur | halo"};
    gpu::TraceRange range {p < VKRH_TILED_PHASES ? names[p] : "tiled: ?"};
    switch (p) {
      case 0:
        f.run(VKRH_STAGE_DOWNSAMPLE);
        start_gather(VKRH_GATHER_HIZ);
        if (!by_request()) start_gather(VKRH_GATHER_ALBEDO);
        else hit_local_rows(compute);
        break;
      case 1:
        if (local_first()) {
          // local rows first: prologue, pinned steps and as much of every march as this rank's own pyramid rows allow run while
          // the depth all-gather is still on the wire; a ray that needs more is parked (csrc/ssr.hip, k_sssr_trace<.., LOCAL>).
          // The TAA, which used to be what ran ahead of the gather, moves behind GTAO: there it covers the hit-colour round.
          f.run(VKRH_STAGE_SSR_TRACE_HEAD);
          break;
        }
        taa_and_halo();
        break;
      case 2:
        wait(VKRH_GATHER_HIZ);
        f.run(VKRH_STAGE_HIZ_TAIL | (local_first() ? VKRH_STAGE_SSR_TRACE_RESUME : VKRH_STAGE_SSR_TRACE));
        if (by_request()) {
          if (cfg.comm) hit_exchange_native();  // count -> all-gather of the counts -> host copy, all on the exchange stream
          else hit_count(compute);
        }
        break;
      case 3:  // GTAO needs the trace's (occlusion, pdf) but not the albedo: it runs ahead of the reference's order to hide the second gather
        finish_halo(VKRH_HALO_AO);
        f.run(VKRH_STAGE_GTAO);
        // first frame (no capacities yet): the host waits for the counts here — GTAO is queued on the device, nothing else yet on the exchange stream
        if (by_request() && cfg.comm && !hit.speculative) hit_exchange_complete();
        copy_halo(VKRH_HALO_AO, true);
        start_halo(VKRH_HALO_AO);
        if (local_first()) taa_and_halo();
        break;
      case 4:
        if (by_request() && !cfg.comm && hit.counted) throw std::runtime_error {"vkrh_tiled_phase: the harness must complete the hit-colour exchange before phase 4"};
        if (by_request() && cfg.comm) hit_exchange_complete();  // the round went out on capacities: the counts arrived long ago, look at them
        wait(VKRH_GATHER_ALBEDO);
        finish_halo(VKRH_HALO_SSR);
        f.run(VKRH_STAGE_SSR_RESOLVE);
        copy_halo(VKRH_HALO_SSR, true);
        start_halo(VKRH_HALO_SSR);
        f.end_frame(false);
        break;
      default: throw std::runtime_error {"vkrh_tiled_phase: phases are 0..4"};
    }
  }
  void taa_and_halo() {
    finish_halo(VKRH_HALO_TAA);
    frame->run(VKRH_STAGE_TAA);
    copy_halo(VKRH_HALO_TAA, true);
    start_halo(VKRH_HALO_TAA);
  }
  // Two frames in flight.  The depth pyramid is the one surface every rank needs whole, its all-gather the longest exchange of
  // the frame, and in the plain order only the TAA runs between the downsample that produces it and the trace that needs it.
  // Here the downsample of frame f + 1 (into the G-buffer's second set) and its all-gather start right after the trace of
  // frame f — the last reader of the whole-frame pyramid — and travel while GTAO, TAA, resolve and blur of frame f run; the TAA
  // moves behind GTAO, where it covers the hit-colour round.  Every step still runs every pass once.  Needs the next frame's
  // G-buffer to be resident when this frame's trace has run (the benchmark's is static; a renderer double-buffers it).
  void pipelined_step() {
    PostFxFrame& f = *frame;
    if (!primed) {  // the first frame's own downsample and gather
      f.run(VKRH_STAGE_DOWNSAMPLE_NEXT);
      start_gather(VKRH_GATHER_HIZ, true);
      f.gbuffer.swap_sets(f.graph);
      hit_local_rows(compute);
      primed = true;
    }
    wait(VKRH_GATHER_HIZ);
    f.run(VKRH_STAGE_HIZ_TAIL | VKRH_STAGE_SSR_TRACE);
    hit_exchange_native();
    f.run(VKRH_STAGE_DOWNSAMPLE_NEXT);          // frame f + 1
    start_gather(VKRH_GATHER_HIZ, true);
    finish_halo(VKRH_HALO_AO);
    f.run(VKRH_STAGE_GTAO);
    if (!hit.speculative) hit_exchange_complete();  // first frame: the host round trip, with GTAO queued
    copy_halo(VKRH_HALO_AO, true);
    start_halo(VKRH_HALO_AO);
    taa_and_halo();
    hit_exchange_complete();
    wait(VKRH_GATHER_ALBEDO);
    finish_halo(VKRH_HALO_SSR);
    f.run(VKRH_STAGE_SSR_RESOLVE);
    copy_halo(VKRH_HALO_SSR, true);
    start_halo(VKRH_HALO_SSR);
    f.end_frame(false);
    f.gbuffer.swap_sets(f.graph);               // what was downsampled for frame f + 1 becomes the current set
    hit_local_rows(compute);                    // its own rows of the whole-frame albedo / normals (after this frame's resolve has read them)
  }
  void step() {
    if (!tiled) { frame->run(VKRH_STAGE_CHAIN); frame->end_frame(false); return; }
    if (!cfg.comm && cfg.world > 1) throw std::runtime_error {"vkrh_tiled_step: no communicator (drive vkrh_tiled_phase from a harness instead)"};
    if (pipeline_enabled && cfg.comm) { pipelined_step(); return; }
    if (!cfg.comm) {  // one rank, no wire: the gathers degenerate to copies of the tile into the frame images
      for (uint32_t p = 0; p < VKRH_TILED_PHASES; p++) {
        if (p == 2) local_gather(VKRH_GATHER_HIZ);
        if (p == 4) local_gather(VKRH_GATHER_ALBEDO);
        phase(p);
      }
      return;
    }
    for (uint32_t p = 0; p < VKRH_TILED_PHASES; p++) phase(p);
  }
  void local_gather(int which) {
    vkr_gather_part p[8];
    const uint32_t n = gather_parts(which, p);
    for (uint32_t i = 0; i < n; i++)
      check(hipMemcpyAsync((uint8_t*)p[i].recv + gather_offsets[which][i][cfg.rank], p[i].send, p[i].bytes, hipMemcpyDeviceToDevice, compute), "local gather");
  }
  void flush() { for (int s = 0; s < 3; s++) finish_halo(s); }
};

template <typename F> int guarded(F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_error = e.what(); return 1; }
  catch (...) { g_error = "unknown exception"; return 2; }
}

// the frame behind a vkrh_* handle; a NULL handle is an error message, not a crash
PostFxFrame& frame_ref(void* frame) {
  if (!frame) throw std::runtime_error{"NULL frame"};
  return *(PostFxFrame*)frame;
}

}  // namespace

extern "C" {

void vkrh_set_allocator(vkrh_alloc_fn alloc, vkrh_free_fn free_fn, void* user) {
  gpu::set_device_allocator((gpu::AllocFn)alloc, (gpu::FreeFn)free_fn, user);
}

void* vkrh_create(const vkrh_config* cfg) {
  PostFxFrame* f = nullptr;
  int rc = guarded([&] {
    if (!cfg) throw std::runtime_error{"vkrh_create: NULL config"};
    if ((cfg->width | cfg->height | (uint32_t)cfg->origin_x | (uint32_t)cfg->origin_y) & 1u)
      throw std::runtime_error{"vkrh_create: window origin and extent must be even"};
    f = new PostFxFrame(*cfg);
  });
  return rc == 0 ? f : nullptr;
}
void vkrh_destroy(void* frame) { delete (PostFxFrame*)frame; }
const char* vkrh_last_error(void) { return g_error.c_str(); }
int vkrh_has_program(const char* name) {
  gpu::register_hot_path_programs();
  return name && gpu::has_program(name) ? 1 : 0;
}

int vkrh_set_camera(void* frame, const vkrh_camera* cam) {
  return guarded([&] { if (!frame || !cam) throw std::runtime_error{"NULL argument"}; frame_ref(frame).set_camera(*cam); });
}
int vkrh_pin_randoms(void* frame, float jitter, uint32_t gtao_frame_count, uint32_t ssr_counter) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    f->gtao.pin_angle_jitter(jitter);
    f->gtao.set_frame_count(gtao_frame_count);
    f->ssr.set_counter(ssr_counter);
  });
}
int vkrh_load_scene(void* frame, const vkr_raster_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                    const vkrh_scene_draw* draws, uint32_t draw_count, const vkrh_scene_texture* textures, uint32_t texture_count) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    if (!f || !vertices || !indices || (!draws && draw_count) || (!textures && texture_count)) throw std::runtime_error{"NULL argument"};
    std::vector<scene::FlatDraw> flat(draw_count);
    for (uint32_t i = 0; i < draw_count; i++) {
      std::memcpy(&flat[i].transform, draws[i].transform, 64);
      flat[i].vertex_offset = draws[i].vertex_offset; flat[i].index_offset = draws[i].index_offset; flat[i].index_count = draws[i].index_count;
      flat[i].albedo_tex_index = draws[i].albedo_tex_index; flat[i].metalic_roughness_index = draws[i].metalic_roughness_index;
      flat[i].clip_alpha = draws[i].clip_alpha != 0;
    }
    std::vector<scene::TextureData> tex(texture_count);
    for (uint32_t i = 0; i < texture_count; i++) {
      if (textures[i].mip_levels == 0 || textures[i].mip_levels > VKR_MAX_MIPS) throw std::runtime_error{"vkrh_load_scene: bad mip count"};
      tex[i].width = textures[i].width; tex[i].height = textures[i].height; tex[i].mip_levels = textures[i].mip_levels;
      for (uint32_t m = 0; m < textures[i].mip_levels; m++) tex[i].levels[m] = textures[i].levels[m];
    }
    f->scene_renderer.reset();
    f->loaded_scene.reset(new scene::CompiledScene(scene::make_scene((const scene::Vertex*)vertices, vertex_count, indices, index_count,
                                                                     flat.data(), draw_count, tex.data(), texture_count)));
    f->scene_renderer.reset(new SceneRenderer(*f->loaded_scene));
    f->scene_renderer->init_pipeline(f->graph, f->gbuffer);  // main.cpp:256-259
    f->scene_renderer->update_scene();
  });
}
int vkrh_pin_screen_trace(void* frame, float angle_jitter, float random_offset, uint32_t frame_count) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    f->screen_trace.pin_randoms(angle_jitter, random_offset);
    f->screen_trace.set_frame_count(frame_count);
  });
}
int vkrh_set_gtao_mode(void* frame, uint32_t use_mis, uint32_t two_directions) {
  return guarded([&] { auto* f = &frame_ref(frame); f->gtao.set_mis(use_mis != 0); f->gtao.set_two_directions(two_directions != 0); });
}
int vkrh_set_synth_flags(void* frame, uint32_t flags) {
  return guarded([&] {
    if (!frame) throw std::runtime_error{"NULL argument"};
    if (flags & ~uint32_t(VKR_SYNTH_TEXTURED_ROUGHNESS)) throw std::runtime_error{"vkrh_set_synth_flags: unknown flag"};
    frame_ref(frame).synth.set_material_flags(flags);
  });
}
int vkrh_set_gathered_mips(void* frame, uint32_t mips) {
  return guarded([&] {
    if (mips < 1 || mips > 4) throw std::runtime_error{"vkrh_set_gathered_mips: 1..4"};
    frame_ref(frame).hiz_gathered_mips = mips;
  });
}
int vkrh_run(void* frame, uint32_t stage_mask) { return guarded([&] { frame_ref(frame).run(stage_mask); }); }
int vkrh_end_frame(void* frame, uint32_t swap_depth) { return guarded([&] { frame_ref(frame).end_frame(swap_depth != 0); }); }
int vkrh_image(void* frame, const char* name, uint32_t base_mip, uint32_t mip_count, vkr_img* out) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    if (!f || !name || !out) throw std::runtime_error{"NULL argument"};
    auto& img = f->graph.get_image(f->lookup(name));
    if (mip_count == 0) mip_count = img->get_mip_levels() - base_mip;
    *out = img->describe(base_mip, mip_count);
  });
}
int vkrh_read_buffer(void* frame, const char* name, void* dst, uint64_t capacity, uint64_t* bytes) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    if (!f || !name || !dst) throw std::runtime_error{"NULL argument"};
    const std::string n{name};
    rendergraph::BufferResourceId id;
    if (n == "reflective_tiles") id = f->ssr.get_reflective_tiles();
    else if (n == "glossy_tiles") id = f->ssr.get_glossy_tiles();
    else if (n == "reflective_indirect") id = f->ssr.get_reflective_indirect();
    else if (n == "glossy_indirect") id = f->ssr.get_glossy_indirect();
    else throw std::runtime_error{"vkrh_read_buffer: unknown buffer '" + n + "'"};
    auto& buf = f->graph.get_buffer(id);
    const uint64_t size = buf->get_size();
    if (bytes) *bytes = size;
    if (capacity < size) throw std::runtime_error{"vkrh_read_buffer: destination too small"};
    void* stream = f->graph.get_stream();
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
        hipMemcpy(dst, buf->device_ptr(stream), size, hipMemcpyDeviceToHost) != hipSuccess)
      throw std::runtime_error{"vkrh_read_buffer: copy failed"};
  });
}
int vkrh_image_layer(void* frame, const char* name, uint32_t layer, vkr_img* out) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    if (!f || !name || !out) throw std::runtime_error{"NULL argument"};
    *out = f->graph.get_image(f->lookup(name))->describe_layer(layer);
  });
}
int vkrh_capture(void* frame, const char* name, uint32_t mip, uint32_t kind, const char* path) {
  return guarded([&] {
    auto* f = &frame_ref(frame);
    if (!f || !name || !path) throw std::runtime_error{"NULL argument"};
    const ReadBackID id = f->readback.read_image(f->graph, f->lookup(name), 0, mip, 0);
    // the request matures frames_count + 1 submits later, like the reference's fenced frames
    for (uint32_t i = 0; i <= f->graph.get_frames_count() + 1 && !f->readback.is_data_available(id); i++) {
      f->graph.submit();
      f->readback.after_submit(f->graph);
    }
    if (!f->readback.is_data_available(id)) throw std::runtime_error{"readback did not mature"};
    ReadBackData data = f->readback.get_data(id);
    bool ok = true;
    if (kind == 0) {
      if (data.texel_fmt != VK_FORMAT_D24_UNORM_S8_UINT) throw std::runtime_error{"depth CSV capture needs a D24S8 image"};
      write_depth_csv(data, path);
    } else if (kind == 1) {
      if (data.texel_size != 4) throw std::runtime_error{"depth PNG capture needs 4-byte texels"};
      ok = write_depth_png(data, path);
    } else if (kind == 2) {
      if (data.texel_size != 4) throw std::runtime_error{"RGBA PNG capture needs 4-byte texels"};
      ok = write_rgba_png(data, path);
    } else {
      throw std::runtime_error{"unknown capture kind"};
    }
    if (!ok) throw std::runtime_error{std::string{"cannot write "} + path};
  });
}
int vkrh_selftest_writers(const char* dir, uint32_t width, uint32_t height) {
  return guarded([&] {
    if (!dir || !width || !height) throw std::runtime_error{"bad arguments"};
    auto make = [&](VkFormat fmt) {
      ReadBackData d;
      d.width = width; d.height = height; d.texel_fmt = fmt; d.texel_size = 4;
      d.bytes.reset(new uint8_t[size_t(width) * height * 4]);
      return d;
    };
    ReadBackData depth = make(VK_FORMAT_D24_UNORM_S8_UINT), color = make(VK_FORMAT_R8G8B8A8_SRGB);
    for (uint32_t y = 0; y < height; y++)
      for (uint32_t x = 0; x < width; x++) {
        const size_t i = size_t(y) * width + x;
        reinterpret_cast<uint32_t*>(depth.bytes.get())[i] = x * 65537u + y * 257u + 0xAB000000u;
        uint8_t* c = color.bytes.get() + 4 * i;
        c[0] = uint8_t(x); c[1] = uint8_t(y); c[2] = uint8_t(x ^ y); c[3] = 7;
      }
    const std::string base{dir};
    write_depth_csv(depth, base + "/depth.csv");
    if (!write_depth_png(depth, base + "/depth.png") || !write_rgba_png(color, base + "/color.png"))
      throw std::runtime_error{"cannot write PNG files under " + base};
  });
}
int vkrh_enable_task_timing(void* frame, uint32_t on) { return guarded([&] { frame_ref(frame).graph.enable_task_timing(on != 0); }); }
int vkrh_enable_task_timing_only(void* frame, const char* task) { return guarded([&] { frame_ref(frame).graph.enable_task_timing(true, task ? task : ""); }); }
const char* vkrh_collect_task_times(void* frame) {
  if (!frame) { g_error = "NULL frame"; return nullptr; }
  auto* f = (PostFxFrame*)frame;
  f->task_names.clear();
  int rc = guarded([&] {
    for (auto& t : f->graph.collect_task_times())
      f->task_names += t.name + " " + std::to_string(t.total_ms) + " " + std::to_string(t.launches) + "\n";
  });
  return rc == 0 ? f->task_names.c_str() : nullptr;
}
int vkrh_selftest_errors(char* buf, uint32_t buf_size) {
  std::string out;
  auto expect = [&](const char* name, std::function<void()> f) {
    try { f(); out += std::string{name} + ": no error\n"; }
    catch (const std::exception& e) { out += std::string{name} + ": " + e.what() + "\n"; }
  };
  expect("unknown_program", [] { gpu::create_compute_pipeline("no_such_program"); });
  expect("known_program", [] { gpu::create_compute_pipeline("gtao_compute_main"); });
  expect("single_mip_depth", [] {
    rendergraph::RenderGraph g;
    const auto u = VK_IMAGE_USAGE_SAMPLED_BIT;
    auto d = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_D24_UNORM_S8_UINT, VK_IMAGE_ASPECT_DEPTH_BIT, 16, 16}, VK_IMAGE_TILING_OPTIMAL, u);
    auto n = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, 16, 16}, VK_IMAGE_TILING_OPTIMAL, u);
    auto v = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 16, 16}, VK_IMAGE_TILING_OPTIMAL, u);
    auto on = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, u);
    auto ov = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, u);
    DownsamplePass pass;
    pass.run(g, n, v, d, on, ov);
  });
  expect("mismatched_outputs", [] {
    rendergraph::RenderGraph g;
    const auto u = VK_IMAGE_USAGE_SAMPLED_BIT;
    auto d = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_D24_UNORM_S8_UINT, VK_IMAGE_ASPECT_DEPTH_BIT, 16, 16, 1, 5, 1}, VK_IMAGE_TILING_OPTIMAL, u);
    auto n = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, 16, 16}, VK_IMAGE_TILING_OPTIMAL, u);
    auto v = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 16, 16}, VK_IMAGE_TILING_OPTIMAL, u);
    auto on = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_UNORM, VK_IMAGE_ASPECT_COLOR_BIT, 4, 8}, VK_IMAGE_TILING_OPTIMAL, u);
    auto ov = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16G16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, u);
    DownsamplePass pass;
    pass.run(g, n, v, d, on, ov);
  });
  expect("incompatible_usage", [] {
    rendergraph::RenderGraph g;
    auto img = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_STORAGE_BIT);
    struct D { rendergraph::ImageViewId a, b; };
    g.add_task<D>("bad",
      [&](D& d, rendergraph::RenderGraphBuilder& b) { d.a = b.sample_image(img, VK_SHADER_STAGE_COMPUTE_BIT); d.b = b.use_storage_image(img, VK_SHADER_STAGE_COMPUTE_BIT, 0, 0); },
      [](D&, rendergraph::RenderResources&, gpu::CmdContext&) {});
  });
  expect("read_then_write_in_separate_tasks", [] {
    rendergraph::RenderGraph g;
    auto img = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_STORAGE_BIT);
    struct D { rendergraph::ImageViewId a; };
    g.add_task<D>("r", [&](D& d, rendergraph::RenderGraphBuilder& b) { d.a = b.sample_image(img, VK_SHADER_STAGE_COMPUTE_BIT); }, [](D&, rendergraph::RenderResources&, gpu::CmdContext&) {});
    g.add_task<D>("w", [&](D& d, rendergraph::RenderGraphBuilder& b) { d.a = b.use_storage_image(img, VK_SHADER_STAGE_COMPUTE_BIT, 0, 0); }, [](D&, rendergraph::RenderResources&, gpu::CmdContext&) {});
  });
  expect("remap_keeps_ids_valid", [&] {
    rendergraph::RenderGraph g;
    auto a = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 8, 8}, VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_STORAGE_BIT);
    auto b = g.create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R16_SFLOAT, VK_IMAGE_ASPECT_COLOR_BIT, 16, 8}, VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_STORAGE_BIT);
    void* pa = g.get_image(a)->device_ptr();
    void* pb = g.get_image(b)->device_ptr();
    g.remap(a, b);
    if (g.get_image(a)->device_ptr() != pb || g.get_image(b)->device_ptr() != pa || g.get_descriptor(a).width != 16)
      throw std::runtime_error{"remap did not swap the table entries"};
  });
  expect("tasks_run_in_submission_order", [&] {
    rendergraph::RenderGraph g;
    std::string order;
    struct D {};
    for (const char* n : {"a", "b", "c"})
      g.add_task<D>(n, [](D&, rendergraph::RenderGraphBuilder&) {}, [&order, n](D&, rendergraph::RenderResources&, gpu::CmdContext&) { order += n; });
    if (!order.empty()) throw std::runtime_error{"run callback executed before submit"};
    g.submit();
    if (order != "abc") throw std::runtime_error{"tasks reordered: " + order};
  });
  expect("ray_query_gtao", [] { rendergraph::RenderGraph g; GTAO gtao{g, 64, 64, true}; });
  expect("ubo_ring_overflow", [] {
    gpu::UniformBufferPool pool;
    struct Big { char b[6000]; };
    pool.allocate_ubo<Big>(); pool.allocate_ubo<Big>(); pool.allocate_ubo<Big>();
  });
  if (buf && buf_size) { std::snprintf(buf, buf_size, "%s", out.c_str()); }
  return 0;
}

// ---- tiled frame ----------------------------------------------------------------------------------------------------
void* vkrh_tiled_create(const vkrh_tiled_config* cfg) {
  TiledFrame* t = nullptr;
  int rc = guarded([&] {
    if (!cfg) throw std::runtime_error{"vkrh_tiled_create: NULL config"};
    t = new TiledFrame(*cfg);
  });
  return rc == 0 ? t : nullptr;
}
void vkrh_tiled_destroy(void* tiled) { delete (TiledFrame*)tiled; }
void* vkrh_tiled_frame(void* tiled) { return tiled ? ((TiledFrame*)tiled)->frame.get() : nullptr; }
// (a NULL handle is an error message, not a crash: the Python harness keeps None for frames that are not native)
static TiledFrame& tiled_ref(void* tiled, const char* what) {
  if (!tiled) throw std::runtime_error{std::string{what} + ": NULL tiled frame"};
  return *(TiledFrame*)tiled;
}
int vkrh_tiled_step(void* tiled) { return guarded([&] { tiled_ref(tiled, "vkrh_tiled_step").step(); }); }
int vkrh_tiled_flush(void* tiled) { return guarded([&] { tiled_ref(tiled, "vkrh_tiled_flush").flush(); }); }
int vkrh_tiled_phase(void* tiled, uint32_t phase) { return guarded([&] { tiled_ref(tiled, "vkrh_tiled_phase").phase(phase); }); }
int vkrh_tiled_gather_parts(void* tiled, uint32_t which, vkr_gather_part* out, uint32_t capacity, uint32_t* count) {
  return guarded([&] {
    if (!tiled || !out || !count || capacity < 8 || which > 1) throw std::runtime_error{"vkrh_tiled_gather_parts: bad arguments (capacity >= 8)"};
    *count = ((TiledFrame*)tiled)->gather_parts((int)which, out);
  });
}
int vkrh_tiled_halo_peers(void* tiled, uint32_t surface, vkr_halo_peer* out, uint32_t capacity, uint32_t* count) {
  return guarded([&] {
    if (!tiled || !out || !count || capacity < 2 || surface > 2) throw std::runtime_error{"vkrh_tiled_halo_peers: bad arguments (capacity >= 2)"};
    *count = ((TiledFrame*)tiled)->halo_peers((int)surface, out);
  });
}
int vkrh_tiled_hit_counts(void* tiled, uint32_t* row) {
  return guarded([&] {
    auto* t = (TiledFrame*)tiled;
    if (!t || !row || !t->by_request()) throw std::runtime_error{"vkrh_tiled_hit_counts: not a tiled frame with hit-colour requests"};
    TiledFrame::check(hipMemcpyAsync(row, t->hit.counts, sizeof(uint32_t) * t->cfg.world, hipMemcpyDeviceToHost, t->compute), "counts");
    TiledFrame::check(hipStreamSynchronize(t->compute), "synchronize");
  });
}
int vkrh_tiled_hit_requests(void* tiled, const uint32_t* matrix, vkr_halo_peer* peers, uint32_t capacity, uint32_t* count) {
  return guarded([&] {
    auto* t = (TiledFrame*)tiled;
    if (!t || !matrix || !peers || !count || capacity < TiledFrame::HIT_PEERS || !t->by_request()) throw std::runtime_error{"vkrh_tiled_hit_requests: bad arguments (capacity >= 16)"};
    *count = t->hit_write(matrix, t->compute, peers);
  });
}
int vkrh_tiled_hit_replies(void* tiled, vkr_halo_peer* peers, uint32_t capacity, uint32_t* count) {
  return guarded([&] {
    auto* t = (TiledFrame*)tiled;
    if (!t || !peers || !count || capacity < TiledFrame::HIT_PEERS || !t->by_request()) throw std::runtime_error{"vkrh_tiled_hit_replies: bad arguments (capacity >= 16)"};
    *count = t->hit_reply(t->compute, peers);
  });
}
int vkrh_tiled_hit_finish(void* tiled) {
  return guarded([&] {
    auto* t = (TiledFrame*)tiled;
    if (!t || !t->by_request()) throw std::runtime_error{"vkrh_tiled_hit_finish: not a tiled frame with hit-colour requests"};
    t->hit_scatter(t->compute);
  });
}
int vkrh_tiled_hit_errors(void* tiled, uint32_t* errors) {
  return guarded([&] {
    auto* t = (TiledFrame*)tiled;
    if (!t || !errors || !t->by_request()) throw std::runtime_error{"vkrh_tiled_hit_errors: not a tiled frame with hit-colour requests"};
    TiledFrame::check(hipDeviceSynchronize(), "synchronize");
    TiledFrame::check(hipMemcpy(errors, t->hit.counts + TiledFrame::HIT_ERRORS, sizeof(uint32_t), hipMemcpyDeviceToHost), "errors");
  });
}
int vkrh_tiled_hit_bytes(void* tiled, uint64_t* bytes) {
  return guarded([&] {
    if (!tiled || !bytes) throw std::runtime_error{"vkrh_tiled_hit_bytes: NULL argument"};
    *bytes = ((TiledFrame*)tiled)->hit.wire_bytes;
  });
}
int vkrh_tiled_hit_rounds(void* tiled, uint64_t* rounds3) {
  return guarded([&] {
    if (!tiled || !rounds3) throw std::runtime_error{"vkrh_tiled_hit_rounds: NULL argument"};
    const auto& h = ((TiledFrame*)tiled)->hit;
    rounds3[0] = h.rounds_speculative; rounds3[1] = h.rounds_exact; rounds3[2] = h.rounds_repeated;
  });
}
int vkrh_hit_capacities(const uint32_t* counts, uint32_t world, uint32_t percent, uint32_t* capacities) {
  return guarded([&] {
    if (!counts || !capacities || world == 0 || world > 16 || percent == 0) throw std::runtime_error{"vkrh_hit_capacities: bad arguments"};
    TiledFrame::hit_capacities(counts, world, percent, capacities);
  });
}
int vkrh_tiled_emulate_wire(void* tiled, void* comm, const uint32_t* counts) {
  return guarded([&] { tiled_ref(tiled, "vkrh_tiled_emulate_wire").emulate_wire(comm, counts); });
}
int vkrh_tiled_pipelined(void* tiled) { return tiled && ((TiledFrame*)tiled)->pipeline_enabled ? 1 : 0; }
int vkrh_tiled_local_first(void* tiled) { return tiled && ((TiledFrame*)tiled)->local_first() ? 1 : 0; }
int vkrh_tiled_time_waits(void* tiled, uint32_t on) {
  return guarded([&] {
    auto* t = &tiled_ref(tiled, "vkrh_tiled_time_waits");
    t->time_waits = on != 0;
    if (!on) {  // marks nobody collected: their events must not pile up
      TiledFrame::check(hipStreamSynchronize(t->compute), "synchronize");
      t->drop_wait_marks();
    }
  });
}
int vkrh_tiled_wait_times(void* tiled, float* ms5) {
  return guarded([&] {
    if (!tiled || !ms5) throw std::runtime_error{"vkrh_tiled_wait_times: NULL argument"};
    ((TiledFrame*)tiled)->collect_waits(ms5);
  });
}
int vkrh_balance_rows(const float* ms, const uint32_t* bounds_in, uint32_t world, uint32_t align, uint32_t min_rows, uint32_t* bounds_out) {
  return guarded([&] {
    if (!ms || !bounds_in || !bounds_out || world == 0 || align == 0) throw std::runtime_error{"vkrh_balance_rows: bad arguments"};
    const uint32_t H = bounds_in[world];
    min_rows = std::max((min_rows + align - 1) / align * align, align);
    if (bounds_in[0] != 0 || H % align || uint64_t(min_rows) * world > H) throw std::runtime_error{"vkrh_balance_rows: the frame does not hold `world` strips of min_rows"};
    double total = 0.0;
    for (uint32_t r = 0; r < world; r++) {
      if (bounds_in[r + 1] <= bounds_in[r] || !(ms[r] > 0.0f)) throw std::runtime_error{"vkrh_balance_rows: bounds must increase and times be positive"};
      total += ms[r];
    }
    // row y of the frame where the cumulative cost (piecewise linear: uniform inside a measured strip) reaches `target`
    auto row_at = [&](double target) {
      double acc = 0.0;
      for (uint32_t r = 0; r < world; r++) {
        if (acc + ms[r] >= target || r + 1 == world)
          return bounds_in[r] + (target - acc) / ms[r] * double(bounds_in[r + 1] - bounds_in[r]);
        acc += ms[r];
      }
      return double(H);
    };
    bounds_out[0] = 0; bounds_out[world] = H;
    for (uint32_t r = 1; r < world; r++) {
      const double y = row_at(total * r / world);
      uint32_t b = uint32_t(y / align + 0.5) * align;
      // keep every strip, the ones still to come included, at least min_rows high
      b = std::max(b, bounds_out[r - 1] + min_rows);
      b = std::min(b, H - (world - r) * min_rows);
      bounds_out[r] = b;
    }
  });
}
const char* vkrh_last_tasks(void* frame) { return frame ? ((PostFxFrame*)frame)->task_names.c_str() : ""; }
const char* vkrh_last_lanes(void* frame) { return frame ? ((PostFxFrame*)frame)->task_lanes.c_str() : ""; }
int vkrh_set_async(void* frame, uint32_t on) { return guarded([&] { frame_ref(frame).graph.set_async(on != 0); }); }

}  // extern "C"
