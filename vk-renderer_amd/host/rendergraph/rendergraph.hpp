// rendergraph/rendergraph.hpp — the reference's task-graph API (src/rendergraph/rendergraph.hpp:17-158,
// resources.hpp:49-114) over HIP streams.
//
// What is kept: ids are 32-bit indices into a table the graph owns; `add_task` runs `create_cb`
// immediately (it declares resource use through the builder) and stores `run_cb`; `submit()` records
// the stored callbacks in submission order, never reordered (rendergraph.cpp:291-305); `remap`
// swaps two table entries so ids stay valid (resources.cpp:89-91); misuse throws
// std::runtime_error.  Usage declarations are checked like the reference's: one task may not read and
// write the same subresource ("Incompatible image usage in task", resources.cpp:350-352).
//
// Barriers: the reference turns the declared usages into Vulkan barriers between *dependent* tasks only, so
// on its single queue independent dispatches overlap (the tail of one pass runs under the head of the next).
// A HIP stream would serialise every kernel instead, so submit() spreads the tasks of one submission over up
// to MAX_LANES streams: a task continues the lane whose last task it depends on (read-after-write,
// write-after-read, write-after-write per image mip / buffer, from the same declarations), otherwise it opens
// a free lane; dependencies across lanes become hipStreamWaitEvent.  Every submit() is a fork / join region
// on the graph's own stream, so everything recorded on that stream before and after (uploads, collectives,
// read-backs, the next submission) is ordered as with one stream.  Off by default (set_async): on this path every
// pass is VALU-bound, co-running kernels split the CUs and the frame got 4 % slower.
#ifndef VKR_HOST_RENDERGRAPH_HPP_INCLUDED
#define VKR_HOST_RENDERGRAPH_HPP_INCLUDED

#include <cinttypes>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../gpu/gpu.hpp"

namespace rendergraph {

struct GraphResources;

struct ImageResourceId {
  ImageResourceId() {}
  uint32_t get_index() const { return index; }
  bool operator==(const ImageResourceId& id) const { return index == id.index; }
 private:
  uint32_t index = ~0u;
  friend struct GraphResources;
};
struct BufferResourceId {
  BufferResourceId() {}
  uint32_t get_index() const { return index; }
  bool operator==(const BufferResourceId& id) const { return index == id.index; }
 private:
  uint32_t index = ~0u;
  friend struct GraphResources;
};
struct ImageViewId {
  ImageViewId(ImageResourceId id, gpu::ImageViewRange view) : res_id{id}, range{view} {}
  ImageViewId() {}
  ImageResourceId get_id() const { return res_id; }
  const gpu::ImageViewRange& get_range() const { return range; }
  operator ImageResourceId() const { return res_id; }
 private:
  ImageResourceId res_id;
  gpu::ImageViewRange range;
};

enum class Usage : uint8_t { None, Sampled, Storage, ColorAttachment, DepthAttachment, TransferRead, TransferWrite };

struct GraphResources {
  ImageResourceId create_image(const gpu::ImageInfo& info, const gpu::FrameWindow& window);
  BufferResourceId create_buffer(VmaMemoryUsage mem, uint64_t size, VkBufferUsageFlags usage);
  void remap(ImageResourceId src, ImageResourceId dst);
  gpu::ImagePtr& get_image(ImageResourceId id);
  const gpu::ImagePtr& get_image(ImageResourceId id) const;
  gpu::BufferPtr& get_buffer(BufferResourceId id);
  // usage tracking of the task being recorded; `log` receives (resource key, write?) for submit()'s lane assignment
  struct Access { uint64_t key; bool write; };  // key: image index << 8 | mip, or 1 << 63 | buffer index
  void declare(ImageResourceId id, uint32_t base_mip, uint32_t mips, Usage usage, uint32_t task_index, std::vector<Access>* log);
  static void declare(BufferResourceId id, bool write, std::vector<Access>* log) { if (log) log->push_back({(1ull << 63) | id.get_index(), write}); }
  size_t image_count() const { return images.size(); }
 private:
  struct Entry {
    gpu::ImagePtr image;
    std::vector<std::pair<uint32_t, Usage>> last;  // per mip: (task index, usage)
  };
  std::vector<Entry> images;
  std::vector<gpu::BufferPtr> buffers;
};

struct RenderGraph;

struct RenderGraphBuilder {  // rendergraph.hpp:17-55
  RenderGraphBuilder(GraphResources& res, uint32_t task, std::vector<GraphResources::Access>* access_log = nullptr)
      : resources{res}, task_index{task}, log{access_log} {}
  ImageViewId use_color_attachment(ImageResourceId id, uint32_t mip, uint32_t layer);
  ImageViewId use_depth_attachment(ImageResourceId id, uint32_t mip, uint32_t layer);
  ImageViewId use_storage_image(ImageResourceId id, VkShaderStageFlags stages, uint32_t mip, uint32_t layer);
  ImageViewId use_storage_image_array(ImageResourceId id, VkShaderStageFlags stages);
  ImageViewId sample_image(ImageResourceId id, VkShaderStageFlags stages, VkImageAspectFlags aspect, uint32_t base_mip,
                           uint32_t mip_count, uint32_t base_layer, uint32_t layer_count);
  ImageViewId sample_image(ImageResourceId id, VkShaderStageFlags stages, VkImageAspectFlags aspect = 0);
  void use_uniform_buffer(BufferResourceId id, VkShaderStageFlags) { GraphResources::declare(id, false, log); }
  void use_storage_buffer(BufferResourceId id, VkShaderStageFlags, bool readonly = true) { GraphResources::declare(id, !readonly, log); }
  void use_indirect_buffer(BufferResourceId id) { GraphResources::declare(id, false, log); }
  void transfer_write(BufferResourceId id) { GraphResources::declare(id, true, log); }
  void transfer_read(BufferResourceId id) { GraphResources::declare(id, false, log); }
  void transfer_read(ImageResourceId id, uint32_t base_mip, uint32_t mip_count, uint32_t base_layer, uint32_t layer_count);
  void transfer_write(ImageResourceId id, uint32_t base_mip, uint32_t mip_count, uint32_t base_layer, uint32_t layer_count);
  gpu::ImageInfo get_image_info(ImageResourceId id);
  // rendergraph.hpp:139: the swapchain image of the current frame.  Headless: an RGBA8_SRGB image of the window's
  // extent, created the first time somebody asks (only DeferedShadingPass's constructor does, for its format).
  ImageResourceId get_backbuffer();

  uint32_t get_frames_count() const { return 1; }
 private:
  GraphResources& resources;
  uint32_t task_index;
  std::vector<GraphResources::Access>* log;
};

struct RenderResources {  // rendergraph.hpp:57-83
  RenderResources(GraphResources& res, gpu::CmdContext& c) : resources{res}, cmd{c} {}
  gpu::BufferPtr& get_buffer(BufferResourceId id) { return resources.get_buffer(id); }
  gpu::ImagePtr& get_image(ImageResourceId id) { return resources.get_image(id); }
  VkImageView get_view(const ImageViewId& ref);
  gpu::ImageViewObject get_image_range(const ImageViewId& ref) { return *(const gpu::ImageViewObject*)get_view(ref); }
  VkDescriptorSet allocate_set(VkDescriptorSetLayout) { return cmd.allocate_set(); }
  VkDescriptorSet allocate_set(const gpu::GraphicsPipeline&, uint32_t) { return cmd.allocate_set(); }
  VkDescriptorSet allocate_set(const gpu::ComputePipeline&, uint32_t) { return cmd.allocate_set(); }
  // rendergraph.hpp:139: the swapchain image of the current frame.  Headless: an RGBA8_SRGB image of the window's
  // extent, created the first time somebody asks (only DeferedShadingPass's constructor does, for its format).
  ImageResourceId get_backbuffer();

  uint32_t get_frames_count() const { return 1; }
  uint32_t get_frame_index() const { return 0; }
  void reset() { views.clear(); }
 private:
  GraphResources& resources;
  gpu::CmdContext& cmd;
  std::vector<std::unique_ptr<gpu::ImageViewObject>> views;
};

struct BaseTask {
  BaseTask(const std::string& task_name) : name{task_name} {}
  virtual void write_commands(RenderResources&, gpu::CmdContext&) = 0;
  virtual ~BaseTask() {}
  const std::string& get_name() const { return name; }
  std::string name;
  std::vector<GraphResources::Access> accesses;  // what create_cb declared
};
template <typename TaskData> using TaskRunCB = std::function<void(TaskData&, RenderResources&, gpu::CmdContext&)>;
template <typename TaskData> using TaskCreateCB = std::function<void(TaskData&, RenderGraphBuilder&)>;
template <typename TaskData> struct Task : BaseTask {
  Task(const std::string& name) : BaseTask{name} {}
  TaskData data;
  TaskRunCB<TaskData> callback;
  void write_commands(RenderResources& resources, gpu::CmdContext& cmd) override { callback(data, resources, cmd); }
};

struct RenderGraph {  // rendergraph.hpp:112-158
  // `stream`: the HIP stream every task is recorded on (nullptr = the default stream)
  explicit RenderGraph(void* stream = nullptr);
  ~RenderGraph();

  template <typename TaskData>
  void add_task(const std::string& name, TaskCreateCB<TaskData> create_cb, TaskRunCB<TaskData> run_cb) {
    std::unique_ptr<Task<TaskData>> ptr{new Task<TaskData>{name}};
    RenderGraphBuilder builder{resources, (uint32_t)tasks.size() + task_base, &ptr->accesses};
    create_cb(ptr->data, builder);
    ptr->callback = run_cb;
    tasks.push_back(std::move(ptr));
  }

  void submit();

  ImageResourceId create_image(VkImageType type, const gpu::ImageInfo& info, VkImageTiling tiling, VkImageUsageFlags usage);
  BufferResourceId create_buffer(VmaMemoryUsage mem, uint64_t size, VkBufferUsageFlags usage) { return resources.create_buffer(mem, size, usage); }
  gpu::ImageInfo get_descriptor(ImageResourceId id) const { return resources.get_image(id)->get_info(); }
  void remap(ImageResourceId src, ImageResourceId dst) { resources.remap(src, dst); }

  // rendergraph.hpp:139: the swapchain image of the current frame.  Headless: an RGBA8_SRGB image of the window's
  // extent, created the first time somebody asks (only DeferedShadingPass's constructor does, for its format).
  ImageResourceId get_backbuffer();

  uint32_t get_frames_count() const { return 1; }
  uint32_t get_frame_index() const { return 0; }

  // ---- additions for headless / tiled use (not in the reference) ----------------------------------
  void set_stream(void* stream) { main_stream = stream; cmd.set_stream(stream); }
  void* get_stream() const { return main_stream; }
  // false (default): every task on the graph's stream (one in-order stream).  true: independent tasks of one
  // submission run on up to MAX_LANES streams, see the header comment.
  void set_async(bool on) { async = on; }
  static constexpr uint32_t MAX_LANES = 3;
  // lane each task of the last submission ran on (tests / diagnostics)
  const std::vector<uint32_t>& last_submitted_lanes() const { return submitted_lanes; }
  // Multi-GPU: this process holds the window (origin, win) of a (full) frame.  Images created at
  // win >> k inherit origin >> k / full >> k; any other extent (LUTs ...) is a standalone image.
  void set_frame_window(uint32_t full_w, uint32_t full_h, int32_t origin_x, int32_t origin_y, uint32_t win_w, uint32_t win_h);
  // standalone image that always covers the whole frame at full >> k (gathered Hi-Z pyramid, ...)
  ImageResourceId create_frame_image(const gpu::ImageInfo& info);
  gpu::ImagePtr& get_image(ImageResourceId id) { return resources.get_image(id); }
  gpu::BufferPtr& get_buffer(BufferResourceId id) { return resources.get_buffer(id); }
  const std::vector<std::string>& last_submitted_tasks() const { return submitted_names; }
  // Per-task device timing: HIP events recorded around every task on the graph's stream (the
  // counterpart of the reference's per-task debug labels, rendergraph.cpp:289-304).  Events are
  // only recorded, never waited on, inside submit(); collect_task_times() synchronises.
  // `only`: time just the task of that name (an event pair costs ~3.5 us of queue time, so timing all nine passes
  // of a 1 ms frame slows it by 7 %); empty = every task.
  void enable_task_timing(bool on, const std::string& only = std::string{});
  struct TaskTime { std::string name; double total_ms = 0; uint32_t launches = 0; };
  std::vector<TaskTime> collect_task_times();

 private:
  GraphResources resources;
  gpu::CmdContext cmd;
  std::vector<std::unique_ptr<BaseTask>> tasks;
  std::vector<std::string> submitted_names;
  uint32_t task_base = 1;
  bool timing = false;
  std::string timing_only;
  struct TimedTask { std::string name; void* start; void* stop; };
  std::vector<TimedTask> timed;
  std::vector<void*> event_pool;
  void* get_event();
  void* main_stream = nullptr;
  bool async = false;
  void* lane_streams[MAX_LANES] = {nullptr, nullptr, nullptr};  // [0] unused: lane 0 is the graph's stream
  std::vector<void*> sync_events;  // one per task of a submission + the fork event, reused by every submission
  std::vector<uint32_t> submitted_lanes;
  bool has_window = false;
  uint32_t full_w = 0, full_h = 0, win_w = 0, win_h = 0;
  ImageResourceId backbuffer;
  bool has_backbuffer = false;
  int32_t org_x = 0, org_y = 0;
};

}  // namespace rendergraph
#endif
