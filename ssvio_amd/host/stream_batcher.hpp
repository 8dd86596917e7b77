// ssvio_amd/host/stream_batcher.hpp -- S independent streams (one System each: BASELINE configs[4], "8 concurrent KITTI streams,
// full frontend + backend") sharing ONE GPU through BATCHED compute calls.
//
// A single stream is a chain of small dependent launches (System::RunStep -> TrackLastFrame -> EstimateCurrentPose,
// /root/reference/src/ssvio/frontend.cpp:100-300, system.cpp:46-52): 0.4 - 0.55 ms per frame of which the GPU is busy for a
// fraction; S streams that each issue their own launches (round 4's `--streams`) share the launch path and four hardware queues
// and reach 4.7 k frames/s at S = 8.  Here every stream still runs its own unmodified System on its own thread, but its two
// per-frame compute calls -- the temporal LK (Compute::TrackLK, temporal) and the pose-only LM (Compute::PoseOnly) -- and its
// window optimisations (BaWindow::Solve) do not go to the GPU one by one: the stream files a request and sleeps; a dispatcher
// thread waits until no stream is running host code any more (every live stream is asleep on a request, or busy in a keyframe's
// own calls), gathers the pending requests of one kind and issues them as ONE library call
//     ssx_lk_track_batch / ssx_pose_only_opt_batch / ssx_ba_window_solve_batch        (include/ssx.h)
// whose per-job results are, bit for bit, those of the single calls -- so every stream's trajectory is byte-identical to its
// single-stream run, whatever S and whatever the interleaving (tests/test_host_gpu.py).  The keyframe path of a stream (masked
// detection, stereo LK, triangulation: ssx_orb_detect_boxes_batch / ssx_lk_track_batch / ssx_triangulate_batch) is batched by a
// second dispatcher on contexts of its own, beside the per-frame batches; a stream's images are read by the GPU from the stream's
// pinned buffers (filled by the stream's own thread), nothing is staged inside the batched calls.
//
// Dispatch order: pose-only before LK.  A stream that comes back from a keyframe is half a frame out of phase with the cohort; with
// the pose-only batch served first the cohort arrives at its next LK request while the straggler still waits there, and they merge.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "compute.hpp"

namespace ssx::host {

class StreamBatcher {
 public:
  // cohorts > 1: the streams are dealt to that many independent batchers (stream k -> cohort k mod cohorts), each with its own
  // dispatchers and contexts: one cohort's batched call runs on the GPU while the other cohort's streams run their host code
  StreamBatcher(int device, int streams, int cohorts = 1);
  ~StreamBatcher();
  StreamBatcher(const StreamBatcher&) = delete;
  StreamBatcher& operator=(const StreamBatcher&) = delete;

  // the Compute of stream k (0 <= k < streams); call once per stream, from any thread, before the stream starts.  The stream's
  // thread must call Finish(k) when it has run its last frame (or died): the dispatcher stops waiting for it.
  std::unique_ptr<Compute> MakeCompute(int k);
  void Finish(int k);

  struct Stats {
    long lk_calls = 0, lk_jobs = 0, po_calls = 0, po_jobs = 0, ba_calls = 0, ba_jobs = 0, kf_calls = 0, kf_jobs = 0;   // kf: detection, stereo LK, triangulation
    double lk_s = 0, po_s = 0, ba_s = 0, kf_s = 0;  // seconds inside the batched library calls
    double wait_s = 0;                               // seconds the per-frame dispatcher waited for the streams' host code
  };
  Stats stats();
  void ResetStats();                                 // (after the streams' warm-up: the counters then describe the frames alone)

  struct Impl;

 private:
  std::vector<std::unique_ptr<Impl>> impls_;
  int n_streams_ = 0;
};

}  // namespace ssx::host
