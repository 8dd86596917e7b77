// ssvio_amd/host/dataset.hpp -- sequence input of the headless runner.
//   LoadKittiImagesTimestamps   common::LoadKittiImagesTimestamps (/root/reference/include/common/read_kitii_dataset.hpp:16-60):
//                               <sequence>/times.txt (one timestamp per non-empty line) and
//                               <sequence>/image_0|image_1/%06d.png for the left / right camera
//   imread_gray                 cv::imread(path, cv::IMREAD_GRAYSCALE) as test_system.cpp:40-41 calls it, for the
//                               files KITTI's gray odometry set holds: non-interlaced grey PNG (8 or 16 bit, with or
//                               without alpha).  Colour / palette / interlaced files are refused with a message.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ssx::host {

struct Image {
  int rows = 0, cols = 0;
  std::vector<uint8_t> data;              // row-major, stride = cols
  uint64_t id = 0;                        // unique per loaded image (lets the compute layer recognise a resident pyramid)
  bool empty() const { return data.empty(); }
  const uint8_t* ptr() const { return data.data(); }
};
using ImagePtr = std::shared_ptr<const Image>;

void LoadKittiImagesTimestamps(const std::string& path_to_sequence, std::vector<std::string>& left_paths,
                               std::vector<std::string>& right_paths, std::vector<double>& timestamps);

// throws std::runtime_error with the reason; an unreadable file gives an empty image like cv::imread
ImagePtr imread_gray(const std::string& path);
// decode from memory (the tests feed hand-built files)
ImagePtr decode_png_gray(const uint8_t* bytes, size_t size);

// Decodes the stereo pairs of a sequence ahead of the tracker on `threads` worker threads (a PNG pair costs several
// milliseconds of inflate, a tracked frame well under one on the GPU) and hands them out in order.  At most `depth`
// decoded pairs are held.  A decode error is rethrown by the Next() call of that frame.
class StereoPrefetcher {
 public:
  struct Pair { ImagePtr left, right; };
  StereoPrefetcher(std::vector<std::string> left_paths, std::vector<std::string> right_paths, size_t count, int threads = 4, size_t depth = 16);
  ~StereoPrefetcher();
  StereoPrefetcher(const StereoPrefetcher&) = delete;
  StereoPrefetcher& operator=(const StereoPrefetcher&) = delete;
  Pair Next();                            // frame 0, 1, 2, ... ; throws std::out_of_range past the end

 private:
  struct Slot { Pair pair; std::exception_ptr error; bool ready = false; };
  void Work();
  std::vector<std::string> left_, right_;
  size_t count_, depth_;
  std::vector<Slot> slots_;               // ring of `depth_` slots: frame i lives in slot i % depth_
  size_t next_claim_ = 0, next_out_ = 0;
  bool stop_ = false;
  std::mutex m_;
  std::condition_variable cv_work_, cv_ready_;
  std::vector<std::thread> workers_;
};

}  // namespace ssx::host
