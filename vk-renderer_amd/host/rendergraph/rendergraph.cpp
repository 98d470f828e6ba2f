// rendergraph/rendergraph.cpp — see rendergraph.hpp.  Reference: src/rendergraph/rendergraph.cpp:271-337
// (submit), resources.cpp:45-125 (resource table, remap), :294-365 (usage tracking).
#include "rendergraph.hpp"

#include <hip/hip_runtime_api.h>

#include <map>
#include <stdexcept>

namespace rendergraph {

ImageResourceId GraphResources::create_image(const gpu::ImageInfo& info, const gpu::FrameWindow& window) {
  Entry e;
  e.image = std::make_shared<gpu::Image>(info, window);
  e.last.assign(info.mip_levels, {0u, Usage::None});
  images.push_back(std::move(e));
  ImageResourceId id;
  id.index = (uint32_t)images.size() - 1;
  return id;
}
BufferResourceId GraphResources::create_buffer(VmaMemoryUsage mem, uint64_t size, VkBufferUsageFlags usage) {
  buffers.push_back(gpu::create_buffer(mem, size, usage));
  BufferResourceId id;
  id.index = (uint32_t)buffers.size() - 1;
  return id;
}
void GraphResources::remap(ImageResourceId src, ImageResourceId dst) { std::swap(images.at(src.index), images.at(dst.index)); }
gpu::ImagePtr& GraphResources::get_image(ImageResourceId id) { return images.at(id.index).image; }
const gpu::ImagePtr& GraphResources::get_image(ImageResourceId id) const { return images.at(id.index).image; }
gpu::BufferPtr& GraphResources::get_buffer(BufferResourceId id) { return buffers.at(id.index); }

static bool is_write(Usage u) { return u == Usage::Storage || u == Usage::ColorAttachment || u == Usage::DepthAttachment || u == Usage::TransferWrite; }

void GraphResources::declare(ImageResourceId id, uint32_t base_mip, uint32_t mips, Usage usage, uint32_t task_index, std::vector<Access>* log) {
  Entry& e = images.at(id.index);
  if (base_mip + mips > e.last.size()) throw std::runtime_error{"Image view outside the mip chain"};
  for (uint32_t m = base_mip; m < base_mip + mips; m++) {
    if (log) log->push_back({((uint64_t)id.index << 8) | m, is_write(usage)});
    auto& [task, prev] = e.last[m];
    if (task == task_index && prev != usage && (is_write(prev) || is_write(usage)))
      throw std::runtime_error{"Incompatible image usage in task"};
    task = task_index;
    prev = usage;
  }
}

// ---- builder ---------------------------------------------------------------------------------------
static gpu::ImageViewRange make_range(VkImageAspectFlags aspect, uint32_t base_mip, uint32_t mips, uint32_t base_layer, uint32_t layers) {
  gpu::ImageViewRange r;
  r.aspect = aspect; r.base_mip = base_mip; r.mips_count = mips; r.base_layer = base_layer; r.layers_count = layers;
  return r;
}
ImageViewId RenderGraphBuilder::use_color_attachment(ImageResourceId id, uint32_t mip, uint32_t layer) {
  resources.declare(id, mip, 1, Usage::ColorAttachment, task_index, log);
  return {id, make_range(VK_IMAGE_ASPECT_COLOR_BIT, mip, 1, layer, 1)};
}
ImageViewId RenderGraphBuilder::use_depth_attachment(ImageResourceId id, uint32_t mip, uint32_t layer) {
  resources.declare(id, mip, 1, Usage::DepthAttachment, task_index, log);
  return {id, make_range(VK_IMAGE_ASPECT_DEPTH_BIT, mip, 1, layer, 1)};
}
ImageViewId RenderGraphBuilder::use_storage_image(ImageResourceId id, VkShaderStageFlags, uint32_t mip, uint32_t layer) {
  resources.declare(id, mip, 1, Usage::Storage, task_index, log);
  return {id, make_range(resources.get_image(id)->get_info().aspect, mip, 1, layer, 1)};
}
ImageViewId RenderGraphBuilder::use_storage_image_array(ImageResourceId id, VkShaderStageFlags) {
  const auto& info = resources.get_image(id)->get_info();
  resources.declare(id, 0, info.mip_levels, Usage::Storage, task_index, log);
  auto r = make_range(info.aspect, 0, info.mip_levels, 0, info.array_layers);
  r.type = VK_IMAGE_VIEW_TYPE_2D_ARRAY;
  return {id, r};
}
ImageViewId RenderGraphBuilder::sample_image(ImageResourceId id, VkShaderStageFlags, VkImageAspectFlags aspect, uint32_t base_mip,
                                             uint32_t mip_count, uint32_t base_layer, uint32_t layer_count) {
  resources.declare(id, base_mip, mip_count, Usage::Sampled, task_index, log);
  return {id, make_range(aspect, base_mip, mip_count, base_layer, layer_count)};
}
ImageViewId RenderGraphBuilder::sample_image(ImageResourceId id, VkShaderStageFlags stages, VkImageAspectFlags aspect) {
  const auto& info = resources.get_image(id)->get_info();
  return sample_image(id, stages, aspect ? aspect : info.aspect, 0, info.mip_levels, 0, info.array_layers);
}
void RenderGraphBuilder::transfer_read(ImageResourceId id, uint32_t base_mip, uint32_t mip_count, uint32_t, uint32_t) {
  resources.declare(id, base_mip, mip_count, Usage::TransferRead, task_index, log);
}
void RenderGraphBuilder::transfer_write(ImageResourceId id, uint32_t base_mip, uint32_t mip_count, uint32_t, uint32_t) {
  resources.declare(id, base_mip, mip_count, Usage::TransferWrite, task_index, log);
}
gpu::ImageInfo RenderGraphBuilder::get_image_info(ImageResourceId id) { return resources.get_image(id)->get_info(); }

VkImageView RenderResources::get_view(const ImageViewId& ref) {
  views.emplace_back(new gpu::ImageViewObject{resources.get_image(ref.get_id()).get(), ref.get_range()});
  return (VkImageView)views.back().get();
}

// ---- graph ---------------------------------------------------------------------------------------------
RenderGraph::RenderGraph(void* stream) : cmd{stream}, main_stream{stream} { gpu::register_hot_path_programs(); }
ImageResourceId RenderGraph::get_backbuffer() {
  if (!has_backbuffer) {
    backbuffer = create_image(VK_IMAGE_TYPE_2D, gpu::ImageInfo{VK_FORMAT_R8G8B8A8_SRGB, VK_IMAGE_ASPECT_COLOR_BIT, win_w ? win_w : 1u, win_h ? win_h : 1u},
                              VK_IMAGE_TILING_OPTIMAL, VK_IMAGE_USAGE_COLOR_ATTACHMENT_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT);
    has_backbuffer = true;
  }
  return backbuffer;
}
RenderGraph::~RenderGraph() {
  for (auto& t : timed) { event_pool.push_back(t.start); event_pool.push_back(t.stop); }
  for (void* e : event_pool) (void)hipEventDestroy((hipEvent_t)e);
  for (void* e : sync_events) (void)hipEventDestroy((hipEvent_t)e);
  for (uint32_t l = 1; l < MAX_LANES; l++)
    if (lane_streams[l]) (void)hipStreamDestroy((hipStream_t)lane_streams[l]);
}

void RenderGraph::enable_task_timing(bool on, const std::string& only) { timing = on; timing_only = only; }
void* RenderGraph::get_event() {
  if (!event_pool.empty()) { void* e = event_pool.back(); event_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) throw std::runtime_error{"hipEventCreate failed"};
  return e;
}
std::vector<RenderGraph::TaskTime> RenderGraph::collect_task_times() {
  std::vector<TaskTime> out;
  std::map<std::string, size_t> index;
  for (auto& t : timed) {
    if (hipEventSynchronize((hipEvent_t)t.stop) != hipSuccess) throw std::runtime_error{"hipEventSynchronize failed"};
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)t.start, (hipEvent_t)t.stop) != hipSuccess) throw std::runtime_error{"hipEventElapsedTime failed"};
    auto it = index.find(t.name);
    if (it == index.end()) { it = index.emplace(t.name, out.size()).first; out.push_back(TaskTime{t.name, 0.0, 0}); }
    out[it->second].total_ms += ms;
    out[it->second].launches += 1;
    event_pool.push_back(t.start);
    event_pool.push_back(t.stop);
  }
  timed.clear();
  return out;
}

void RenderGraph::set_frame_window(uint32_t fw, uint32_t fh, int32_t ox, int32_t oy, uint32_t ww, uint32_t wh) {
  if (ox < 0 || oy < 0 || (uint32_t)ox + ww > fw || (uint32_t)oy + wh > fh) throw std::runtime_error{"Frame window outside the frame"};
  has_window = true;
  full_w = fw; full_h = fh; org_x = ox; org_y = oy; win_w = ww; win_h = wh;
}

ImageResourceId RenderGraph::create_image(VkImageType type, const gpu::ImageInfo& info, VkImageTiling, VkImageUsageFlags) {
  if (type != VK_IMAGE_TYPE_2D) throw std::runtime_error{"Only 2D images exist on this path"};
  gpu::FrameWindow w;
  if (has_window) {
    for (uint32_t k = 0; k < 2; k++) {
      if (info.width == (win_w >> k) && info.height == (win_h >> k)) {
        w.full_width = full_w >> k; w.full_height = full_h >> k;
        w.origin_x = org_x >> k; w.origin_y = org_y >> k;
        break;
      }
    }
  }
  return resources.create_image(info, w);
}
ImageResourceId RenderGraph::create_frame_image(const gpu::ImageInfo& info) { return resources.create_image(info, gpu::FrameWindow{}); }

static void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error{std::string{what} + ": " + hipGetErrorString(e)};
}

void RenderGraph::submit() {
  // rendergraph.cpp:291-305: tasks are recorded in submission order.  Lane assignment and cross-lane waits follow
  // the hazards between the tasks' declared accesses (see the header comment); lane 0 is the graph's own stream.
  cmd.begin();
  RenderResources res{resources, cmd};
  submitted_names.clear();
  submitted_lanes.clear();
  std::vector<std::unique_ptr<BaseTask>> run;
  run.swap(tasks);
  task_base += (uint32_t)run.size();

  struct Hazard { int writer = -1; std::vector<int> readers; };
  std::map<uint64_t, Hazard> hazards;
  std::vector<uint32_t> lane_of(run.size(), 0);
  int lane_tail[MAX_LANES];
  bool lane_forked[MAX_LANES];
  for (uint32_t l = 0; l < MAX_LANES; l++) { lane_tail[l] = -1; lane_forked[l] = l == 0; }
  const bool spread = async && run.size() > 1;
  if (spread) {
    while (sync_events.size() < run.size() + 1) {
      hipEvent_t e;
      hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
      sync_events.push_back(e);
    }
    hip_check(hipEventRecord((hipEvent_t)sync_events[run.size()], (hipStream_t)main_stream), "hipEventRecord");  // fork point
  }
  auto stream_of = [&](uint32_t lane) -> void* {
    if (lane == 0) return main_stream;
    if (!lane_streams[lane]) {
      int least = 0, greatest = 0;  // side lanes: lowest priority the device offers (gfx950: same as the default stream)
      hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
      hipStream_t s;
      hip_check(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least), "hipStreamCreateWithPriority");
      lane_streams[lane] = s;
    }
    return lane_streams[lane];
  };

  for (size_t i = 0; i < run.size(); i++) {
    auto& t = run[i];
    uint32_t lane = 0;
    std::vector<int> deps;
    if (spread) {
      for (const auto& a : t->accesses) {
        const Hazard& h = hazards[a.key];
        if (h.writer >= 0) deps.push_back(h.writer);
        if (a.write) deps.insert(deps.end(), h.readers.begin(), h.readers.end());
      }
      // continue the lane whose last task this one depends on (the latest such task); else open a free lane
      int best = -1;
      for (uint32_t l = 0; l < MAX_LANES; l++)
        for (int d : deps)
          if (lane_tail[l] == d && d > best) { best = d; lane = l; }
      if (best < 0) {
        lane = 0;
        for (uint32_t l = 0; l < MAX_LANES; l++)
          if (lane_tail[l] < 0) { lane = l; break; }
      }
      void* stream = stream_of(lane);
      if (!lane_forked[lane]) {  // nothing of this submission may start before what the graph's stream held at submit()
        hip_check(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)sync_events[run.size()], 0), "hipStreamWaitEvent");
        lane_forked[lane] = true;
      }
      for (int d : deps)
        if (lane_of[d] != lane) hip_check(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)sync_events[d], 0), "hipStreamWaitEvent");
      cmd.set_stream(stream);
    }
    lane_of[i] = lane;
    lane_tail[lane] = (int)i;
    submitted_names.push_back(t->get_name());
    submitted_lanes.push_back(lane);
    cmd.push_label(t->get_name().c_str());
    if (timing && (timing_only.empty() || timing_only == t->get_name())) {
      TimedTask tt{t->get_name(), get_event(), get_event()};
      (void)hipEventRecord((hipEvent_t)tt.start, (hipStream_t)cmd.get_stream());
      t->write_commands(res, cmd);
      (void)hipEventRecord((hipEvent_t)tt.stop, (hipStream_t)cmd.get_stream());
      timed.push_back(std::move(tt));
    } else {
      t->write_commands(res, cmd);
    }
    cmd.pop_label();
    if (spread) {
      hip_check(hipEventRecord((hipEvent_t)sync_events[i], (hipStream_t)cmd.get_stream()), "hipEventRecord");
      for (const auto& a : t->accesses) {
        Hazard& h = hazards[a.key];
        if (a.write) { h.writer = (int)i; h.readers.clear(); }
        else h.readers.push_back((int)i);
      }
    }
  }
  if (spread) {  // join: the graph's stream continues only after every lane
    for (uint32_t l = 1; l < MAX_LANES; l++)
      if (lane_tail[l] >= 0) hip_check(hipStreamWaitEvent((hipStream_t)main_stream, (hipEvent_t)sync_events[lane_tail[l]], 0), "hipStreamWaitEvent");
    cmd.set_stream(main_stream);
  }
}

}  // namespace rendergraph
