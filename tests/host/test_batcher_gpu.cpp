// tests/host/test_batcher_gpu.cpp -- StreamBatcher (ssvio_amd/host/stream_batcher.hpp) driven directly, without the tracker on top:
// S threads make the keyframe and per-frame compute calls of one frame each (masked detection, stereo LK, triangulation, two chained
// temporal LK calls, pose-only) on their own images; every result must be, byte for byte, what the same calls return one by one
// through SsxCompute.  Then the same round with two streams filing requests the library rejects (a negative point count): those two
// streams get the exception, the other streams of the same batched calls get their results.
//   test_batcher_gpu [streams=6] [cohorts=1]
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../ssvio_amd/host/compute.hpp"
#include "../../ssvio_amd/host/stream_batcher.hpp"

using namespace ssx::host;

static int g_failed = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); ++g_failed; } } while (0)

static const int ROWS = 160, COLS = 256;

// a textured image: three octaves of bilinear value noise; (dx, dy) shifts the texture
static Image make_image(uint32_t seed, double dx, double dy, uint64_t id)
{
  Image im;
  im.rows = ROWS; im.cols = COLS; im.id = id; im.data.resize((size_t)ROWS * COLS);
  auto h = [&](int x, int y, int o) {
    uint32_t v = seed * 2654435761u ^ (uint32_t)(x * 73856093) ^ (uint32_t)(y * 19349663) ^ (uint32_t)(o * 83492791);
    v ^= v >> 13; v *= 0x5bd1e995u; v ^= v >> 15;
    return (double)(v & 0xffff) / 65535.0;
  };
  for (int y = 0; y < ROWS; ++y)
    for (int x = 0; x < COLS; ++x) {
      double acc = 0;
      const double cell[3] = {24.0, 9.0, 3.5}, amp[3] = {90.0, 70.0, 50.0};
      for (int o = 0; o < 3; ++o) {
        const double fx = (x + dx) / cell[o] + 100.0, fy = (y + dy) / cell[o] + 100.0;
        const int ix = (int)std::floor(fx), iy = (int)std::floor(fy);
        const double ax = fx - ix, ay = fy - iy;
        const double v = (1 - ay) * ((1 - ax) * h(ix, iy, o) + ax * h(ix + 1, iy, o)) + ay * ((1 - ax) * h(ix, iy + 1, o) + ax * h(ix + 1, iy + 1, o));
        acc += amp[o] * v;
      }
      im.data[(size_t)y * COLS + x] = (uint8_t)std::min(255.0, std::max(0.0, acc + 20.0));
    }
  return im;
}

struct StreamData {
  Image L0, R0, L1, L2;
  std::vector<int32_t> boxes;
};

struct Results {
  std::vector<ssx_keypoint> kps;
  std::vector<float> right_pts, next1, next2;
  std::vector<uint8_t> st_right, st1, st2, tri_ok, inl;
  std::vector<double> xyz;
  double pose[7];
  int n_inl = -1;
  std::string po_error, tri_error;
  bool operator==(const Results& o) const
  {
    return kps.size() == o.kps.size() && (kps.empty() || std::memcmp(kps.data(), o.kps.data(), kps.size() * sizeof(ssx_keypoint)) == 0) &&
           right_pts.size() == o.right_pts.size() && std::memcmp(right_pts.data(), o.right_pts.data(), right_pts.size() * 4) == 0 && st_right == o.st_right &&
           next1.size() == o.next1.size() && std::memcmp(next1.data(), o.next1.data(), next1.size() * 4) == 0 && st1 == o.st1 &&
           next2.size() == o.next2.size() && std::memcmp(next2.data(), o.next2.data(), next2.size() * 4) == 0 && st2 == o.st2 &&
           xyz.size() == o.xyz.size() && std::memcmp(xyz.data(), o.xyz.data(), xyz.size() * 8) == 0 && tri_ok == o.tri_ok && inl == o.inl &&
           std::memcmp(pose, o.pose, sizeof(pose)) == 0 && n_inl == o.n_inl;
  }
};

enum Fault { NONE, BAD_POSE_ONLY, BAD_TRIANGULATE };

// the calls of one keyframe + two tracked frames, as FrontEnd makes them (frontend.cpp:100-300, 302-544)
static void run_stream(Compute& c, const StreamData& d, Results& r, Fault fault)
{
  ssx_orb_params prm{};
  prm.nfeatures = 150; prm.scale_factor = 1.2f; prm.nlevels = 4; prm.ini_th_fast = 20; prm.min_th_fast = 7;
  c.DetectBoxes(d.L0, d.boxes, prm, r.kps);
  const int n = (int)r.kps.size();
  std::vector<float> pts(2 * (size_t)n);
  for (int i = 0; i < n; ++i) { pts[2 * i] = r.kps[i].x; pts[2 * i + 1] = r.kps[i].y; }
  r.right_pts = pts;
  c.TrackLK(d.L0, d.R0, pts, r.right_pts, r.st_right, false);
  std::vector<double> uvL(2 * (size_t)n), uvR(2 * (size_t)n);
  for (int i = 0; i < 2 * n; ++i) { uvL[i] = pts[i]; uvR[i] = r.right_pts[i]; }
  ssx_stereo_rig rig{};
  rig.fx = 300; rig.fy = 300; rig.cx = COLS / 2.0; rig.cy = ROWS / 2.0; rig.baseline = 0.5;
  r.xyz.assign(3 * (size_t)n, 0.0); r.tri_ok.assign(n, 0);
  try {
    c.Triangulate(fault == BAD_TRIANGULATE ? -1 : n, uvL.data(), uvR.data(), rig, nullptr, r.xyz.data(), r.tri_ok.data());
  } catch (const std::exception& e) {
    r.tri_error = e.what();
  }
  r.next1 = pts;
  c.TrackLK(d.L0, d.L1, pts, r.next1, r.st1, true);
  r.next2 = r.next1;
  c.TrackLK(d.L1, d.L2, r.next1, r.next2, r.st2, true);          // chained: L1's pyramid is resident
  const double K4[4] = {rig.fx, rig.fy, rig.cx, rig.cy};
  const double p0[7] = {0, 0, 0, 1, 0.01, -0.01, 0.02};      // qx qy qz qw tx ty tz
  std::memcpy(r.pose, p0, sizeof(p0));
  std::vector<double> uv(2 * (size_t)n);
  for (int i = 0; i < 2 * n; ++i) uv[i] = r.next2[i];
  r.inl.assign(n, 0);
  try {
    r.n_inl = c.PoseOnly(r.pose, K4, fault == BAD_POSE_ONLY ? -1 : n, r.xyz.data(), uv.data(), r.inl.data());
  } catch (const std::exception& e) {
    r.po_error = e.what();
  }
}

int main(int argc, char** argv)
{
  const int S = argc > 1 ? std::atoi(argv[1]) : 6, C = argc > 2 ? std::atoi(argv[2]) : 1;
  std::vector<StreamData> data(S);
  uint64_t id = 1;
  for (int k = 0; k < S; ++k) {
    StreamData& d = data[k];
    d.L0 = make_image(17 + k, 0, 0, id++);
    d.R0 = make_image(17 + k, 7.0 + k % 3, 0, id++);       // a disparity of 7 - 9 px
    d.L1 = make_image(17 + k, 1.5, 0.5, id++);
    d.L2 = make_image(17 + k, 3.0, 1.0, id++);
    d.boxes = {40 + k, 30, 90 + k, 70, 150, 60 + k, 200, 100};
  }
  // one by one, each stream on a Compute of its own
  std::vector<Results> ref(S);
  for (int k = 0; k < S; ++k) {
    auto c = MakeSsxCompute(0);
    run_stream(*c, data[k], ref[k], NONE);
    CHECK(ref[k].kps.size() > 40 && ref[k].n_inl >= 0 && ref[k].po_error.empty() && ref[k].tri_error.empty());
    int tracked = 0;
    for (uint8_t s_ : ref[k].st2) tracked += s_;
    CHECK(tracked > (int)ref[k].kps.size() / 2);
  }
  for (int round = 0; round < 2; ++round) {
    StreamBatcher batcher(0, S, C);
    std::vector<Results> got(S);
    std::vector<std::string> died(S);
    std::vector<std::unique_ptr<Compute>> comp;
    for (int k = 0; k < S; ++k) comp.push_back(batcher.MakeCompute(k));
    std::vector<std::thread> th;
    for (int k = 0; k < S; ++k)
      th.emplace_back([&, k] {
        const Fault f = round == 0 ? NONE : k == 1 ? BAD_POSE_ONLY : k == S - 2 ? BAD_TRIANGULATE : NONE;
        try { run_stream(*comp[k], data[k], got[k], f); } catch (const std::exception& e) { died[k] = e.what(); }
        batcher.Finish(k);
      });
    for (auto& t : th) t.join();
    for (int k = 0; k < S; ++k) {
      CHECK(died[k].empty());
      if (!died[k].empty()) std::fprintf(stderr, "stream %d: %s\n", k, died[k].c_str());
      const Fault f = round == 0 ? NONE : k == 1 ? BAD_POSE_ONLY : k == S - 2 ? BAD_TRIANGULATE : NONE;
      if (f == NONE) {
        CHECK(got[k] == ref[k]);
      } else if (f == BAD_POSE_ONLY) {
        CHECK(!got[k].po_error.empty() && got[k].next2 == ref[k].next2 && got[k].st2 == ref[k].st2 && got[k].xyz == ref[k].xyz);
      } else {
        CHECK(!got[k].tri_error.empty() && got[k].next2 == ref[k].next2 && got[k].st2 == ref[k].st2 && got[k].po_error.empty());
      }
    }
    const StreamBatcher::Stats st = batcher.stats();
    std::printf("round %d: LK calls %ld (%ld jobs), pose-only %ld (%ld), keyframe calls %ld (%ld)\n", round, st.lk_calls, st.lk_jobs, st.po_calls, st.po_jobs,
                st.kf_calls, st.kf_jobs);
    if (round == 0 && C == 1) CHECK(st.lk_jobs == 2L * S && st.po_jobs == (long)S);
  }
  if (g_failed) { std::fprintf(stderr, "%d checks failed\n", g_failed); return 1; }
  std::printf("test_batcher_gpu: ok (%d streams, %d cohorts)\n", S, C);
  return 0;
}
