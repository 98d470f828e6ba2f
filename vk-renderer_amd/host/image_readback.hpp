// image_readback.hpp — ReadBackSystem, public interface of src/image_readback.hpp:11-52, plus the
// capture writers of src/main.cpp:118-176 (CSV of 24-bit hex depth, PNG of depth words, PNG of RGBA8
// with alpha forced to 255) so that outputs of this build and captures of the Vulkan reference are
// diffable in one format.
//
// HIP design: the "ImageRead" task is one hipMemcpy2DAsync from the pitch-linear image into pinned
// host memory on the graph's stream; the request matures after frames_count + 1 calls of
// after_submit() like the reference's fenced frames (image_readback.cpp:93,118-124), at which point
// the stream is synchronised once for all matured requests.
#ifndef IMAGE_READBACK_HPP_INCLUDED
#define IMAGE_READBACK_HPP_INCLUDED

#include <memory>
#include <string>
#include <unordered_map>

#include "rendergraph/rendergraph.hpp"

struct ReadBackData {
  uint32_t width = 0;
  uint32_t height = 0;
  VkFormat texel_fmt = VK_FORMAT_UNDEFINED;
  uint32_t texel_size = 0;
  std::unique_ptr<uint8_t[]> bytes {nullptr};
};

using ReadBackID = uint64_t;
const ReadBackID INVALID_READBACK = ~0ull;

struct ReadBackSystem {
  ReadBackID read_image(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId image);
  ReadBackID read_image(rendergraph::RenderGraph &graph, rendergraph::ImageResourceId image, VkImageAspectFlags aspect, uint32_t mip, uint32_t layer);

  void after_submit(rendergraph::RenderGraph &graph);
  bool is_data_available(ReadBackID id) const { return processed_requests.count(id); }

  ReadBackData get_data(ReadBackID id);
  void clear();
  ~ReadBackSystem() { clear(); }

private:
  struct Request {
    uint32_t wait_frames;
    uint32_t width;
    uint32_t height;
    VkFormat texel_fmt;
    uint32_t texel_size;
    std::shared_ptr<void> pinned;  // tightly packed rows, hipHostMalloc
  };

  ReadBackID next_request_id = 0;
  std::unordered_map<ReadBackID, Request> requests;
  std::unordered_map<ReadBackID, ReadBackData> processed_requests;
};

// ---- capture writers (main.cpp:118-176); `path` replaces the hard-coded "captures/..." names ----
// "y, 0,1,...\n" header, then one "y,0x<hex24>,...\n" row per image row
void write_depth_csv(const ReadBackData &image, const std::string &path);
// depth words masked to 24 bits, written as a 4-channel PNG (R = low byte)
bool write_depth_png(ReadBackData &image, const std::string &path);
// RGBA8 with alpha forced to 255
bool write_rgba_png(ReadBackData &image, const std::string &path);
// 8-bit PNG encoder used by the two above (stored deflate blocks: byte-exact pixels, no compression)
bool write_png_rgba8(const std::string &path, uint32_t width, uint32_t height, const uint8_t *rgba);

#endif
