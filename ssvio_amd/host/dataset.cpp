// ssvio_amd/host/dataset.cpp -- KITTI sequence listing and the grey-PNG reader (see dataset.hpp)
#include "dataset.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ssx::host {

void LoadKittiImagesTimestamps(const std::string& seq, std::vector<std::string>& left_paths, std::vector<std::string>& right_paths,
                               std::vector<double>& timestamps)
{
  std::ifstream f(seq + "/times.txt");
  if (!f.is_open()) throw std::runtime_error("LoadKittiImagesTimestamps: cannot open " + seq + "/times.txt");
  std::string line;
  while (std::getline(f, line)) {
    if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::stringstream ss(line);
    double t = 0;
    ss >> t;
    timestamps.push_back(t);
  }
  const size_t n = timestamps.size();
  left_paths.resize(n);
  right_paths.resize(n);
  char name[32];
  for (size_t i = 0; i < n; ++i) {
    std::snprintf(name, sizeof(name), "%06zu.png", i);
    left_paths[i] = seq + "/image_0/" + name;
    right_paths[i] = seq + "/image_1/" + name;
  }
}

namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

std::atomic<uint64_t> g_image_id{1};

}  // namespace

ImagePtr decode_png_gray(const uint8_t* b, size_t size)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (size < 8 || std::memcmp(b, sig, 8) != 0) throw std::runtime_error("png: bad signature");
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1;
  std::vector<uint8_t> z;
  size_t pos = 8;
  bool end = false;
  while (!end && pos + 12 <= size) {
    const uint32_t len = be32(b + pos);
    const uint8_t* type = b + pos + 4;
    const uint8_t* body = b + pos + 8;
    if (pos + 12 + (size_t)len > size) throw std::runtime_error("png: truncated chunk");
    if (crc32(crc32(0, type, 4), body, len) != be32(body + len)) throw std::runtime_error("png: chunk CRC mismatch");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error("png: bad IHDR");
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9];
      if (body[10] != 0 || body[11] != 0) throw std::runtime_error("png: unknown compression / filter method");
      if (body[12] != 0) throw std::runtime_error("png: interlaced files are not supported");
    } else if (!std::memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w == 0 || h == 0 || w > 16384 || h > 16384) throw std::runtime_error("png: missing or implausible IHDR");
  if (!((ctype == 0 || ctype == 4) && (depth == 8 || depth == 16)))
    throw std::runtime_error("png: only 8/16-bit grey (colour type 0 or 4) is supported, got type " + std::to_string(ctype) + " depth " +
                             std::to_string(depth));
  const int bpp = (depth / 8) * (ctype == 4 ? 2 : 1);               // bytes per pixel
  const size_t row = (size_t)w * bpp;
  std::vector<uint8_t> raw((row + 1) * h);
  uLongf out_len = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &out_len, z.data(), (uLong)z.size());
  if (zr != Z_OK || out_len != raw.size()) throw std::runtime_error("png: inflate failed or wrong image size");
  // undo the scanline filters in place, one specialised loop per filter type (the per-byte dispatch of a generic loop
  // costs more than the inflate)
  std::vector<uint8_t> zero(row, 0);
  const size_t B = (size_t)bpp;
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* cur = raw.data() + (size_t)y * (row + 1) + 1;
    const uint8_t* up = y ? cur - (row + 1) : zero.data();
    switch (cur[-1]) {
      case 0: break;
      case 1:
        for (size_t i = B; i < row; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - B]);
        break;
      case 2:
        for (size_t i = 0; i < row; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
        break;
      case 3:
        for (size_t i = 0; i < B && i < row; ++i) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));
        for (size_t i = B; i < row; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - B] + up[i]) >> 1));
        break;
      case 4:
        for (size_t i = 0; i < B && i < row; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);               // paeth(0, b, 0) = b
        for (size_t i = B; i < row; ++i) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - B], up[i], up[i - B]));
        break;
      default: throw std::runtime_error("png: unknown filter type");
    }
  }
  auto img = std::make_shared<Image>();
  img->rows = (int)h; img->cols = (int)w;
  img->data.resize((size_t)w * h);
  img->id = g_image_id.fetch_add(1);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* cur = raw.data() + (size_t)y * (row + 1) + 1;
    uint8_t* dst = img->data.data() + (size_t)y * w;
    for (uint32_t x = 0; x < w; ++x) dst[x] = cur[(size_t)x * bpp];   // 16-bit: the high byte (png_set_strip_16); alpha dropped
  }
  return img;
}

ImagePtr imread_gray(const std::string& path)
{
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return std::make_shared<Image>();
  std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return decode_png_gray(bytes.data(), bytes.size());
}

StereoPrefetcher::StereoPrefetcher(std::vector<std::string> left_paths, std::vector<std::string> right_paths, size_t count, int threads,
                                   size_t depth)
    : left_(std::move(left_paths)), right_(std::move(right_paths)), count_(std::min(count, left_.size())), depth_(std::max<size_t>(depth, 1)),
      slots_(depth_)
{
  for (int i = 0; i < std::max(threads, 1); ++i) workers_.emplace_back([this] { Work(); });
}

StereoPrefetcher::~StereoPrefetcher()
{
  {
    std::lock_guard<std::mutex> lk(m_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto& w : workers_) w.join();
}

void StereoPrefetcher::Work()
{
  for (;;) {
    size_t i;
    {
      std::unique_lock<std::mutex> lk(m_);
      // claim the next frame once its ring slot has been handed out
      cv_work_.wait(lk, [this] { return stop_ || (next_claim_ < count_ && next_claim_ < next_out_ + depth_); });
      if (stop_) return;
      i = next_claim_++;
    }
    Pair pair;
    std::exception_ptr error;
    try {
      pair.left = imread_gray(left_[i]);
      pair.right = imread_gray(right_[i]);
    } catch (...) {
      error = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      Slot& s = slots_[i % depth_];
      s.pair = std::move(pair); s.error = error; s.ready = true;
    }
    cv_ready_.notify_all();
  }
}

StereoPrefetcher::Pair StereoPrefetcher::Next()
{
  std::unique_lock<std::mutex> lk(m_);
  if (next_out_ >= count_) throw std::out_of_range("StereoPrefetcher: past the end of the sequence");
  Slot& s = slots_[next_out_ % depth_];
  cv_ready_.wait(lk, [&s] { return s.ready; });
  Pair out = std::move(s.pair);
  std::exception_ptr error = s.error;
  s = Slot();
  ++next_out_;
  lk.unlock();
  cv_work_.notify_all();
  if (error) std::rethrow_exception(error);
  return out;
}

}  // namespace ssx::host
