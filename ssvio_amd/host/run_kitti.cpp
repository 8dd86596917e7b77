// ssvio_amd/host/run_kitti.cpp -- headless equivalent of the reference's test_system
// (/root/reference/test/test_system.cpp:18-53): load a KITTI-layout stereo sequence, feed every pair to System::RunStep,
// save the keyframe trajectory in TUM format.  Same two flags (gflags spelling), plus a frame limit and an output path.
//
//   ssx_run_kitti --config_yaml_path=cfg.yaml --kitti_dataset_path=<sequence dir> [--max_frames=N] [--trajectory=out.txt]
//                 [--device=0] [--decode_threads=8] [--streams=1] [--preload=0] [--batched=0] [--warmup=1]
// The PNG pairs are decoded ahead of the tracker on worker threads (StereoPrefetcher); everything else is the
// reference's single loop.  --streams=K runs K independent copies of the loop in K threads of this process (each with
// its own System, GPU contexts and prefetcher) on the same sequence: a single stream is latency-bound, several fill the
// GPU (BASELINE configs[4] runs one stream per GPU; this is the one-GPU version of it).  --kitti_dataset_path may be a
// comma-separated list: stream k then runs sequence k mod (number of sequences) -- several DIFFERENT sequences at once.
// --batched=C (with --streams=K): the streams' per-frame compute calls and window optimisations go to the GPU as batched library
// calls (StreamBatcher, stream_batcher.hpp), the streams dealt to C cohorts that are batched independently (C = 2: one cohort's call
// on the GPU while the other's streams run their host code); every stream's trajectory stays byte-identical to its single-stream run.
// --warmup=1 (default): System::Warmup before the first frame -- the library's kernels are loaded and its workspaces sized on a
// synthetic frame, outside the frame loop and its clock (a stream's first window solve alone costs 15 - 20 ms cold, 0.5 ms warm);
// the trajectory is the same with --warmup=0.
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "stream_batcher.hpp"
#include "system.hpp"

namespace {

bool flag(const char* arg, const char* name, std::string& out)
{
  std::string a(arg);
  while (!a.empty() && a[0] == '-') a.erase(0, 1);
  const size_t n = std::strlen(name);
  if (a.compare(0, n, name) != 0 || a.size() <= n || a[n] != '=') return false;
  out = a.substr(n + 1);
  return true;
}

}  // namespace

int main(int argc, char** argv)
{
  std::string config, dataset, max_frames_s, trajectory, device_s, threads_s, streams_s, preload_s, batched_s, warmup_s;
  for (int i = 1; i < argc; ++i) {
    if (flag(argv[i], "config_yaml_path", config) || flag(argv[i], "kitti_dataset_path", dataset) || flag(argv[i], "max_frames", max_frames_s) ||
        flag(argv[i], "trajectory", trajectory) || flag(argv[i], "device", device_s) || flag(argv[i], "decode_threads", threads_s) || flag(argv[i], "streams", streams_s) || flag(argv[i], "preload", preload_s) || flag(argv[i], "batched", batched_s) || flag(argv[i], "warmup", warmup_s))
      continue;
    std::fprintf(stderr, "unknown argument %s\n", argv[i]);
    return 2;
  }
  if (config.empty() || dataset.empty()) {
    std::fprintf(stderr, "usage: %s --config_yaml_path=<yaml> --kitti_dataset_path=<sequence dir> [--max_frames=N] [--trajectory=<tum file>] [--device=0] [--decode_threads=8] [--streams=1] [--preload=0]\n",
                 argv[0]);
    return 2;
  }
  using namespace ssx::host;
  using clk = std::chrono::steady_clock;
  try {
    // one sequence, or a comma-separated list of sequences for --streams
    struct Sequence {
      std::vector<std::string> left_paths, right_paths;
      std::vector<double> timestamps;
      size_t num_images = 0;
      std::vector<StereoPrefetcher::Pair> preloaded;
    };
    std::vector<Sequence> seqs;
    for (size_t pos = 0; pos <= dataset.size();) {
      const size_t comma = std::min(dataset.find(',', pos), dataset.size());
      Sequence sq;
      LoadKittiImagesTimestamps(dataset.substr(pos, comma - pos), sq.left_paths, sq.right_paths, sq.timestamps);
      sq.num_images = sq.left_paths.size();
      if (!max_frames_s.empty()) sq.num_images = std::min(sq.num_images, (size_t)std::atol(max_frames_s.c_str()));
      seqs.push_back(std::move(sq));
      pos = comma + 1;
    }
    std::vector<std::string>& left_paths = seqs[0].left_paths;
    std::vector<std::string>& right_paths = seqs[0].right_paths;
    std::vector<double>& timestamps = seqs[0].timestamps;
    const size_t num_images = seqs[0].num_images;
    std::printf("Num Images: %zu\n", num_images);

    const int streams = streams_s.empty() ? 1 : std::max(1, std::atoi(streams_s.c_str()));
    const bool warmup = warmup_s.empty() || std::atoi(warmup_s.c_str()) != 0;
    if (streams > 1) {
      const int device = device_s.empty() ? 0 : std::atoi(device_s.c_str());
      const int dthreads = threads_s.empty() ? 8 : std::atoi(threads_s.c_str());
      std::vector<double> seconds(streams, 0.0);
      std::vector<size_t> keyframes(streams, 0), frames(streams, 0);
      std::vector<std::string> errors(streams);
      // --preload=1: decode every sequence once, before the clock starts (isolates tracking from PNG decoding)
      if (!preload_s.empty() && std::atoi(preload_s.c_str()) != 0)
        for (Sequence& sq : seqs) {
          StereoPrefetcher pf(sq.left_paths, sq.right_paths, sq.num_images, 32);
          for (size_t ni = 0; ni < sq.num_images; ++ni) sq.preloaded.push_back(pf.Next());
        }
      const int cohorts = batched_s.empty() ? 0 : std::atoi(batched_s.c_str());      // --batched=C: C cohorts of streams, each batched on its own
      const bool batched = cohorts > 0;
      std::unique_ptr<StreamBatcher> batcher;
      if (batched) batcher = std::make_unique<StreamBatcher>(device, streams, cohorts);
      // every stream builds its System (GPU contexts, window) first; the clock of the aggregate figure starts when all are ready
      std::mutex bm;
      std::condition_variable bcv;
      int ready = 0;
      clk::time_point t_go;
      std::vector<clk::time_point> t_end(streams);
      std::vector<std::thread> workers;
      const auto t_all0 = clk::now();
      for (int k = 0; k < streams; ++k)
        workers.emplace_back([&, k] {
          struct Fin { StreamBatcher* b; int k; ~Fin() { if (b) b->Finish(k); } } fin{batcher.get(), k};   // the dispatcher must not wait for a stream that is gone
          bool counted = false;
          auto arrive = [&] {
            std::unique_lock<std::mutex> lk(bm);
            if (!counted) { counted = true; if (++ready == streams) { if (batcher) batcher->ResetStats(); t_go = clk::now(); bcv.notify_all(); } }
            return lk;
          };
          try {
            const Sequence& sq = seqs[(size_t)k % seqs.size()];
            const size_t num_images = sq.num_images;
            frames[k] = num_images;
            System sys(config, batcher ? batcher->MakeCompute(k) : nullptr, device);
            std::unique_ptr<StereoPrefetcher> pf;
            if (sq.preloaded.empty()) pf = std::make_unique<StereoPrefetcher>(sq.left_paths, sq.right_paths, num_images, dthreads);
            StereoPrefetcher::Pair first;
            if (num_images > 0) first = pf ? pf->Next() : sq.preloaded[0];
            if (warmup && first.left && !first.left->empty()) sys.Warmup(first.left->rows, first.left->cols);   // (batched: the streams' warm-up calls are batched too)
            {
              auto lk = arrive();
              bcv.wait(lk, [&] { return ready == streams; });
            }
            const auto t0 = clk::now();
            for (size_t ni = 0; ni < num_images; ++ni) {
              StereoPrefetcher::Pair pair = ni == 0 ? first : pf ? pf->Next() : sq.preloaded[ni];
              if (pair.left->empty() || pair.right->empty()) throw std::runtime_error("Failed to load image " + sq.left_paths[ni]);
              sys.RunStep(pair.left, pair.right, sq.timestamps[ni]);
            }
            sys.backend().WaitIdle();
            t_end[k] = clk::now();
            seconds[k] = std::chrono::duration<double>(t_end[k] - t0).count();
            keyframes[k] = sys.map().GetAllKeyFrames().size();
            fin.b = nullptr;
            if (batcher) batcher->Finish(k);                                  // before the trajectory is written and the System is torn down
            if (!trajectory.empty()) sys.SaveTrajectoryTUM(trajectory + "." + std::to_string(k));
          } catch (const std::exception& e) {
            errors[k] = e.what();
            arrive();
          }
        });
      for (auto& w : workers) w.join();
      const double wall = std::chrono::duration<double>(clk::now() - t_all0).count();
      double sum = 0;
      size_t total_frames = 0;
      clk::time_point last = t_go;
      for (int k = 0; k < streams; ++k) {
        if (!errors[k].empty()) { std::fprintf(stderr, "fatal (stream %d): %s\n", k, errors[k].c_str()); return 1; }
        sum += frames[k] / std::max(seconds[k], 1e-9);
        total_frames += frames[k];
        last = std::max(last, t_end[k]);
        if (streams <= 16) std::printf("stream %d: %.1f frames/s, %zu keyframes\n", k, frames[k] / std::max(seconds[k], 1e-9), keyframes[k]);
      }
      const double span = std::chrono::duration<double>(last - t_go).count();
      std::printf("%d streams: aggregate %.1f frames/s (sum of the streams' loops incl. decoding), wall %.2f s incl. context creation\n", streams, sum, wall);
      std::printf("%d streams%s: %zu frames in %.4f s from the common start to the last stream's end = %.1f frames/s\n", streams, batched ? " (batched)" : "",
                  total_frames, span, total_frames / std::max(span, 1e-9));
      if (batcher) {
        const StreamBatcher::Stats bs = batcher->stats();
        std::printf("batched calls: LK %ld (%.1f jobs each), pose-only %ld (%.1f), window solves %ld (%.1f)\n", bs.lk_calls, bs.lk_jobs / std::max(1.0, (double)bs.lk_calls),
                    bs.po_calls, bs.po_jobs / std::max(1.0, (double)bs.po_calls), bs.ba_calls, bs.ba_jobs / std::max(1.0, (double)bs.ba_calls));
        std::printf("batched time: LK %.1f ms (%.3f per call), pose-only %.1f ms (%.3f), window solves %.1f ms (%.3f), keyframe calls (detection, stereo LK, "
                    "triangulation) %ld x %.1f jobs, %.1f ms (%.3f per call), dispatcher waiting for the streams' host code %.1f ms\n",
                    1e3 * bs.lk_s, 1e3 * bs.lk_s / std::max(1L, bs.lk_calls), 1e3 * bs.po_s, 1e3 * bs.po_s / std::max(1L, bs.po_calls), 1e3 * bs.ba_s,
                    1e3 * bs.ba_s / std::max(1L, bs.ba_calls), bs.kf_calls, bs.kf_jobs / std::max(1.0, (double)bs.kf_calls), 1e3 * bs.kf_s,
                    1e3 * bs.kf_s / std::max(1L, bs.kf_calls), 1e3 * bs.wait_s);
      }
      return 0;
    }
    System system(config, nullptr, device_s.empty() ? 0 : std::atoi(device_s.c_str()));
    double t_io = 0, t_step = 0;
    StereoPrefetcher prefetch(left_paths, right_paths, num_images, threads_s.empty() ? 8 : std::atoi(threads_s.c_str()));
    StereoPrefetcher::Pair first;
    if (num_images > 0) first = prefetch.Next();
    if (warmup && first.left && !first.left->empty()) {
      const auto tw = clk::now();
      system.Warmup(first.left->rows, first.left->cols);
      std::printf("warm-up (kernels loaded, workspaces sized; outside the frame loop): %.1f ms\n", std::chrono::duration<double, std::milli>(clk::now() - tw).count());
    }
    const auto t_begin = clk::now();
    for (size_t ni = 0; ni < num_images; ++ni) {
      const auto t0 = clk::now();
      StereoPrefetcher::Pair pair = ni == 0 ? first : prefetch.Next();
      ImagePtr left = pair.left, right = pair.right;
      if (left->empty() || right->empty()) {
        std::fprintf(stderr, "Failed to load image at: %s\n", (left->empty() ? left_paths[ni] : right_paths[ni]).c_str());
        return 1;
      }
      const auto t1 = clk::now();
      system.RunStep(left, right, timestamps[ni]);
      const auto t2 = clk::now();
      t_io += std::chrono::duration<double>(t1 - t0).count();
      t_step += std::chrono::duration<double>(t2 - t1).count();
      if (ni % 100 == 99) std::printf("Has processed %zu frames.\n", ni + 1);
    }
    system.SaveTrajectoryTUM(trajectory);

    const StageTimes& st = system.frontend().times();
    const Backend::Stats& bs = system.backend().stats();
    const char* status[] = {"INITING", "TRACKING_GOOD", "TRACKING_BAD", "LOST"};
    std::printf("frames %zu  keyframes %zu  map points %zu  final status %s\n", num_images, system.map().GetAllKeyFrames().size(),
                system.map().GetAllMapPoints().size(), status[(int)system.frontend().status()]);
    const double t_all = std::chrono::duration<double>(clk::now() - t_begin).count();
    std::printf("RunStep %.3f ms/frame (%.1f frames/s); waiting for decoded images %.3f ms/frame; whole loop %.1f frames/s\n",
                1e3 * t_step / std::max<size_t>(num_images, 1), num_images / std::max(t_step, 1e-9), 1e3 * t_io / std::max<size_t>(num_images, 1),
                num_images / std::max(t_all, 1e-9));
    auto line = [](const char* name, double s, long n) {
      if (n) std::printf("  %-24s %6ld calls  %8.3f ms/call\n", name, n, 1e3 * s / n);
    };
    line("Detect", st.detect, st.n_detect);
    line("LK last -> current", st.lk_temporal, st.n_lk_temporal);
    line("LK left -> right", st.lk_stereo, st.n_lk_stereo);
    line("pose-only LM", st.pose_only, st.n_pose_only);
    line("triangulation", st.triangulate, st.n_triangulate);
    line("keyframe insert + BA", st.bundle_adjust, st.n_bundle_adjust);
    std::printf("  local BA: %ld windows, %ld LM iterations, %ld edges, %ld outlier edges\n", bs.windows, bs.lm_iterations, bs.edges, bs.outlier_edges);
    if (bs.windows)
      std::printf("  per window on the host side: map + window edits %.3f ms, export + solve call %.3f ms, write-back %.3f ms\n", 1e3 * bs.t_insert / bs.windows,
                  1e3 * bs.t_solve / bs.windows, 1e3 * bs.t_apply / bs.windows);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  return 0;
}
