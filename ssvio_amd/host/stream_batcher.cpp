// ssvio_amd/host/stream_batcher.cpp -- see stream_batcher.hpp
#include "stream_batcher.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <stdexcept>

#include "../../include/ssx_shim.hpp"

namespace ssx::host {

namespace {
enum class St { RUNNING, PENDING_LK, PENDING_PO, PENDING_DET, PENDING_LKS, PENDING_TRI, PENDING_BA, INFLIGHT, LONGOP, DONE, IDLE };
}

struct StreamBatcher::Impl {
  // Request slots: k < S is stream k's own thread (one request at a time); S + k is the worker thread of stream k's backend when the
  // backend is asynchronous (Backend.Async: 1 -- it files window solves while the stream's thread goes on tracking).  A backend slot
  // rests in IDLE, which no dispatcher waits for.
  Impl(int device_, int streams) : device(device_), S(streams), lk_ctx(device_), po_ctx(device_), ba_ctx(device_), det_ctx(device_), lks_ctx(device_),
                                   tri_ctx(device_), state(2 * (size_t)streams, St::RUNNING), lk_req(streams), lk_rows(streams, 0), lk_cols(streams, 0),
                                   po_req(streams), det_req(streams), det_prm(streams), tri_req(streams), ba_win(2 * (size_t)streams, nullptr),
                                   ba_res(2 * (size_t)streams, nullptr), error(2 * (size_t)streams)
  {
    n_in[(int)St::RUNNING] = streams;
    n_in[(int)St::IDLE] = streams;
    for (int k = 0; k < streams; ++k) state[(size_t)S + k] = St::IDLE;
    for (int k = 0; k < 2 * streams; ++k) sleeper.push_back(std::make_unique<Sleeper>());
    disp = std::thread([this] { DispatchLoop(); });
    ba_disp = std::thread([this] { KeyframeLoop(); });
  }
  ~Impl()
  {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv_disp.notify_all();
    if (disp.joinable()) disp.join();
    if (ba_disp.joinable()) ba_disp.join();
    ssx_host_free(pin_arena);
  }

  // The streams' pinned image buffers are slices of ONE arena: buffer `which` of stream k at (which * S + k) * slice.  The `next`
  // images of a cohort's LK jobs then lie at a constant distance and cross PCIe in one DMA copy (ssx_lk_track_batch).
  uint8_t* PinSlice(int k, int which, size_t bytes)
  {
    std::lock_guard<std::mutex> lk(pin_mu);
    const size_t need = (bytes + 255) & ~size_t(255);
    if (!pin_arena) {
      pin_slice = need;
      pin_arena = static_cast<uint8_t*>(ssx_host_alloc(pin_slice * 2 * (size_t)S));
      if (!pin_arena) throw std::runtime_error("StreamBatcher: no pinned memory for the streams' images");
    }
    if (need > pin_slice) return nullptr;           // (a stream with larger images than the first one: it keeps buffers of its own)
    return pin_arena + ((size_t)which * S + k) * pin_slice;
  }

  int count(St s) const { return n_in[(int)s]; }
  // (m held) -- the dispatchers' conditions can only BECOME true when the last running stream stops running: they are woken then,
  // not at every one of the S state changes of a round
  bool Move(int k, St s)
  {
    --n_in[(int)state[k]]; ++n_in[(int)s];
    state[k] = s;
    return n_in[(int)St::RUNNING] == 0;
  }

  // the calling stream sleeps until a dispatcher has served its request; throws what the batch call reported.  Every stream sleeps
  // on a condition variable of its own: S streams woken through one shared mutex queue up behind each other (64 wake-ups x a futex
  // round trip each: more than the batched call they waited for).
  void SubmitAndWait(int k, St pending)
  {
    bool wake;
    {
      std::lock_guard<std::mutex> lk(m);
      error[k].clear();
      wake = Move(k, pending);
    }
    if (wake) cv_disp.notify_all();
    Sleeper& sl = *sleeper[k];
    std::unique_lock<std::mutex> lk(sl.mu);
    sl.cv.wait(lk, [&] { return sl.done; });
    sl.done = false;
    if (!error[k].empty()) throw std::runtime_error(error[k]);
  }
  void SetState(int k, St s)
  {
    bool wake;
    {
      std::lock_guard<std::mutex> lk(m);
      wake = Move(k, s);
    }
    if (wake) cv_disp.notify_all();
  }
  // (m held by the caller) the streams of a finished batch run again
  void Release(const std::vector<int>& who, const std::vector<std::string>& errs)
  {
    for (size_t i = 0; i < who.size(); ++i) { error[who[i]] = errs[i]; Move(who[i], who[i] < S ? St::RUNNING : St::IDLE); }
  }
  // One batched call for the requests idx (positions in `who`).  If it fails, every request goes once more in a call of its own:
  // only the stream whose request is at fault sees the error, the others get their results (the library validates a job table before
  // it touches anything, so a rejected call has not run any of its jobs).
  template <class F>
  static void Isolated(const std::vector<size_t>& idx, std::vector<std::string>& errs, long& calls, long& jobs, F&& call)
  {
    try {
      call(idx);
      ++calls; jobs += (long)idx.size();
      return;
    } catch (const std::exception& e) {
      if (idx.size() == 1) { errs[idx[0]] = e.what(); return; }
    }
    for (size_t i : idx) {
      try { call(std::vector<size_t>{i}); ++calls; ++jobs; } catch (const std::exception& e) { errs[i] = e.what(); }
    }
  }
  // the requests of `who` in groups of equal key (image size, extractor settings ...): one call per group
  template <class Same>
  static std::vector<std::vector<size_t>> Groups(size_t n, Same&& same)
  {
    std::vector<std::vector<size_t>> g;
    std::vector<char> taken(n, 0);
    for (size_t a = 0; a < n; ++a) {
      if (taken[a]) continue;
      g.emplace_back();
      for (size_t b = a; b < n; ++b) if (!taken[b] && same(a, b)) { g.back().push_back(b); taken[b] = 1; }
    }
    return g;
  }
  void LkCall(ssx::Context& ctx, const std::vector<int>& who, const std::vector<size_t>& sub)
  {
    std::vector<ssx_lk_job> jobs;
    for (size_t i : sub) jobs.push_back(lk_req[who[i]]);
    ssx_lk_params p;
    ssx_lk_default_params(&p);
    p.win = 11; p.max_level = 3; p.max_iters = 30; p.eps = 0.01; p.use_initial_flow = 1;
    ctx.check(ssx_lk_track_batch(ctx.get(), (int32_t)jobs.size(), jobs.data(), lk_rows[who[sub[0]]], lk_cols[who[sub[0]]], &p, 1));
  }
  void WakeStreams(const std::vector<int>& who)
  {
    for (int k : who) {
      Sleeper& sl = *sleeper[k];
      { std::lock_guard<std::mutex> lk(sl.mu); sl.done = true; }
      sl.cv.notify_one();
    }
  }

  void DispatchLoop()
  {
    std::unique_lock<std::mutex> lk(m);
    std::vector<int> who;
    using clk = std::chrono::steady_clock;
    for (;;) {
      const auto tw0 = clk::now();
      cv_disp.wait(lk, [&] { return quit || (count(St::RUNNING) == 0 && count(St::PENDING_LK) + count(St::PENDING_PO) > 0); });
      if (quit) return;
      st.wait_s += std::chrono::duration<double>(clk::now() - tw0).count();
      // pose-only first: the cohort then reaches its next LK request where a straggler from a keyframe already waits (see the header)
      const St kind = count(St::PENDING_PO) > 0 ? St::PENDING_PO : St::PENDING_LK;
      who.clear();
      for (int k = 0; k < S; ++k) if (state[k] == kind) { who.push_back(k); Move(k, St::INFLIGHT); }
      lk.unlock();
      std::vector<std::string> errs(who.size());
      long d_calls = 0, d_jobs = 0;
      const auto tc0 = clk::now();
      if (kind == St::PENDING_PO) {
        std::vector<size_t> all(who.size());
        for (size_t i = 0; i < all.size(); ++i) all[i] = i;
        Isolated(all, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) {
          std::vector<ssx_pose_only_job> jobs;
          for (size_t i : sub) jobs.push_back(po_req[who[i]]);
          po_ctx.check(ssx_pose_only_opt_batch(po_ctx.get(), (int32_t)jobs.size(), jobs.data()));
        });
      } else {
        // (jobs of one call share the image size: streams of another size go in a call of their own)
        for (const auto& g : Groups(who.size(), [&](size_t a, size_t b) { return lk_rows[who[a]] == lk_rows[who[b]] && lk_cols[who[a]] == lk_cols[who[b]]; }))
          Isolated(g, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) { LkCall(lk_ctx, who, sub); });
      }
      const double dt = std::chrono::duration<double>(clk::now() - tc0).count();
      lk.lock();
      if (kind == St::PENDING_PO) { st.po_calls += d_calls; st.po_jobs += d_jobs; st.po_s += dt; } else { st.lk_calls += d_calls; st.lk_jobs += d_jobs; st.lk_s += dt; }
      Release(who, errs);
      lk.unlock();
      WakeStreams(who);
      lk.lock();
    }
  }

  // The keyframe path of the streams (FrontEnd::DetectFeatures -> FindFeaturesInRight -> TriangulateNewPoints -> Backend: masked
  // detection, stereo LK, triangulation, window optimisation), batched like the per-frame calls, on contexts of its own beside them.
  // The EARLIEST stage that has requests is served first, so that streams which reached their keyframe a little later catch up and
  // all of them meet in one window solve (a batched solve costs what a single one costs).
  void KeyframeLoop()
  {
    std::unique_lock<std::mutex> lk(m);
    std::vector<int> who;
    auto n_kf = [&] { return count(St::PENDING_DET) + count(St::PENDING_LKS) + count(St::PENDING_TRI) + count(St::PENDING_BA); };
    for (;;) {
      // (streams in a call of their own -- LONGOP -- are about to file the next request of this path: wait for them)
      const bool at_rest = cv_disp.wait_for(lk, std::chrono::milliseconds(2), [&] { return quit || (n_kf() > 0 && count(St::RUNNING) == 0 && count(St::LONGOP) == 0); });
      if (quit) return;
      if (!at_rest) {
        // An asynchronous backend's window solve has waited 2 ms for the front-ends to come to rest -- they may be waiting for IT
        // (Backend::WaitIdle at the end of a sequence, the map mutex): it is served with what has gathered so far.
        bool backend_waits = false;
        for (int k = S; k < 2 * S; ++k) backend_waits = backend_waits || state[k] == St::PENDING_BA;
        if (!backend_waits) continue;
      }
      const St kind = count(St::PENDING_DET) > 0 ? St::PENDING_DET : count(St::PENDING_LKS) > 0 ? St::PENDING_LKS : count(St::PENDING_TRI) > 0 ? St::PENDING_TRI
                                                                                                                                              : St::PENDING_BA;
      who.clear();
      for (int k = 0; k < 2 * S; ++k) if (state[k] == kind) { who.push_back(k); Move(k, St::INFLIGHT); }
      lk.unlock();
      cv_disp.notify_all();                         // (the per-frame dispatcher does not wait for streams that are in flight here)
      std::vector<std::string> errs(who.size());
      long d_calls = 0, d_jobs = 0;
      const auto tb0 = std::chrono::steady_clock::now();
      std::vector<size_t> all(who.size());
      for (size_t i = 0; i < all.size(); ++i) all[i] = i;
      if (kind == St::PENDING_BA) {
        Isolated(all, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) {
          std::vector<ssx_ba_window*> wins;
          std::vector<ssx_ba_result> res;
          for (size_t i : sub) { wins.push_back(ba_win[who[i]]); res.push_back(*ba_res[who[i]]); }
          ba_ctx.check(ssx_ba_window_solve_batch((int32_t)wins.size(), wins.data(), res.data()));
          for (size_t j = 0; j < sub.size(); ++j) *ba_res[who[sub[j]]] = res[j];
        });
      } else if (kind == St::PENDING_TRI) {
        Isolated(all, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) {
          std::vector<ssx_triangulate_job> jobs;
          for (size_t i : sub) jobs.push_back(tri_req[who[i]]);
          tri_ctx.check(ssx_triangulate_batch(tri_ctx.get(), (int32_t)jobs.size(), jobs.data()));
        });
      } else if (kind == St::PENDING_LKS) {
        for (const auto& g : Groups(who.size(), [&](size_t a, size_t b) { return lk_rows[who[a]] == lk_rows[who[b]] && lk_cols[who[a]] == lk_cols[who[b]]; }))
          Isolated(g, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) { LkCall(lks_ctx, who, sub); });
      } else {
        // detection: the jobs of a call share image size, stride and extractor settings
        auto same = [&](size_t a, size_t b) {
          const int ka = who[a], kb = who[b];
          return lk_rows[kb] == lk_rows[ka] && lk_cols[kb] == lk_cols[ka] && det_req[kb].stride == det_req[ka].stride &&
                 std::memcmp(&det_prm[kb], &det_prm[ka], sizeof(ssx_orb_params)) == 0;
        };
        for (const auto& g : Groups(who.size(), same))
          Isolated(g, errs, d_calls, d_jobs, [&](const std::vector<size_t>& sub) {
            std::vector<ssx_orb_detect_job> jobs;
            for (size_t i : sub) jobs.push_back(det_req[who[i]]);
            const int k0 = who[sub[0]];
            const ssx_status rc = ssx_orb_detect_boxes_batch(det_ctx.get(), (int32_t)jobs.size(), jobs.data(), lk_rows[k0], lk_cols[k0], &det_prm[k0], 1);
            // (SSX_ERR_CAPACITY is per image -- the other images of the call are complete: every stream looks at its own count)
            if (rc != SSX_ERR_CAPACITY) det_ctx.check(rc);
          });
      }
      (void)d_calls; (void)d_jobs;
      const double dtb = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
      lk.lock();
      if (kind == St::PENDING_BA) { ++st.ba_calls; st.ba_jobs += (long)who.size(); st.ba_s += dtb; }
      else { ++st.kf_calls; st.kf_jobs += (long)who.size(); st.kf_s += dtb; }
      Release(who, errs);
      lk.unlock();
      WakeStreams(who);
      lk.lock();
    }
  }

  std::mutex pin_mu;
  uint8_t* pin_arena = nullptr;
  size_t pin_slice = 0;
  int device, S;
  ssx::Context lk_ctx, po_ctx, ba_ctx, det_ctx, lks_ctx, tri_ctx;
  std::mutex m;
  std::condition_variable cv_disp;
  struct Sleeper { std::mutex mu; std::condition_variable cv; bool done = false; };
  std::vector<std::unique_ptr<Sleeper>> sleeper;
  int n_in[16] = {0};                               // streams per state
  std::vector<St> state;
  std::vector<ssx_lk_job> lk_req;
  std::vector<int> lk_rows, lk_cols;
  std::vector<ssx_pose_only_job> po_req;
  std::vector<ssx_orb_detect_job> det_req;
  std::vector<ssx_orb_params> det_prm;
  std::vector<ssx_triangulate_job> tri_req;
  std::vector<ssx_ba_window*> ba_win;
  std::vector<ssx_ba_result*> ba_res;
  std::vector<std::string> error;
  StreamBatcher::Stats st;
  bool quit = false;
  std::thread disp, ba_disp;
};

namespace {

struct LongOp {                                   // a call the stream makes on its own context (a keyframe's path)
  StreamBatcher::Impl& im; int k;
  LongOp(StreamBatcher::Impl& i, int k_) : im(i), k(k_) { im.SetState(k, St::LONGOP); }
  ~LongOp() { im.SetState(k, St::RUNNING); }
};

class BatchedBaWindow final : public BaWindow {
 public:
  BatchedBaWindow(StreamBatcher::Impl& im, int k, const double* K4, const double* cam_ext14, const ssx_ba_options& opt)
      : im_(im), k_(k), owner_(std::this_thread::get_id())
  {
    im_.ba_ctx.check(ssx_ba_window_create(im_.ba_ctx.get(), &opt, K4, cam_ext14, &win_));
    im_.ba_ctx.check(ssx_ba_window_set_fix_rule(win_, 1));
  }
  ~BatchedBaWindow() override { ssx_ba_window_destroy(win_); }
  void Push(int64_t kf_id, const double* pose7, int n_new, const int64_t* new_ids, const double* new_xyz, const uint8_t* new_fixed, int n_obs,
            const int64_t* obs_lm, const double* obs_uv, const uint8_t* obs_cam) override
  {
    check(ssx_ba_window_push_keyframe(win_, kf_id, pose7, 0, n_new, new_ids, new_xyz, new_fixed, n_obs, obs_lm, obs_uv, obs_cam));
  }
  void Pop(int64_t kf_id) override { check(ssx_ba_window_pop_keyframe(win_, kf_id)); }
  void RemoveLandmarks(int n, const int64_t* lm_ids) override { check(ssx_ba_window_remove_landmarks(win_, n, lm_ids, nullptr)); }
  void RemoveFlagged(int n_obs, const uint8_t* flags) override { check(ssx_ba_window_remove_flagged(win_, n_obs, flags, nullptr)); }
  void Size(int& nk, int& nl, int& no) override
  {
    int32_t a = 0, b = 0, c = 0;
    check(ssx_ba_window_size(win_, &a, &b, &c));
    nk = a; nl = b; no = c;
  }
  void Export(int64_t* kf_ids, int64_t* lm_ids, uint8_t* point_fixed, int32_t* edge_pose, int32_t* edge_point, double* edge_uv) override
  {
    check(ssx_ba_window_export(win_, kf_ids, nullptr, nullptr, lm_ids, nullptr, point_fixed, edge_pose, edge_point, edge_uv, nullptr));
  }
  void Solve(ssx_ba_result& res) override
  {
    // the stream's own thread files the request in the stream's slot; the worker thread of an asynchronous backend has a slot of its
    // own (the stream's thread goes on filing per-frame requests meanwhile)
    const int slot = std::this_thread::get_id() == owner_ ? k_ : im_.S + k_;
    im_.ba_win[slot] = win_; im_.ba_res[slot] = &res;
    im_.SubmitAndWait(slot, St::PENDING_BA);
  }

 private:
  // (these calls only edit the window's host mirror; the shared context's error text may belong to another stream's call, so the
  // message names the status instead)
  void check(ssx_status st) const { if (st != SSX_OK) throw std::runtime_error("ssx_ba_window: edit failed with status " + std::to_string((int)st)); }
  StreamBatcher::Impl& im_;
  int k_;
  std::thread::id owner_;                            // the stream's thread (it makes the window: Backend's constructor)
  ssx_ba_window* win_ = nullptr;
};

class BatchedCompute final : public Compute {
 public:
  BatchedCompute(StreamBatcher::Impl& im, int k) : im_(im), k_(k), frame_(im.device) {}
  ~BatchedCompute() override { if (own_pins_) { ssx_host_free(pin_[0]); ssx_host_free(pin_[1]); } }

  void Detect(const Image& img, const uint8_t* mask, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) override
  {
    owner_ = std::this_thread::get_id();           // (the front-end calls come from the stream's thread)
    LongOp op(im_, k_);
    kps.assign((size_t)prm.nfeatures + 260 + 64, ssx_keypoint{});
    int32_t n = 0;
    ssx_status st = ssx_orb_detect(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, mask, img.cols, &prm, (int32_t)kps.size(), kps.data(), &n);
    if (st == SSX_ERR_CAPACITY && n > (int32_t)kps.size()) {
      kps.assign((size_t)n, ssx_keypoint{});
      st = ssx_orb_detect(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, mask, img.cols, &prm, (int32_t)kps.size(), kps.data(), &n);
    }
    frame_.check(st);
    kps.resize(n);
  }
  void DetectBoxes(const Image& img, const std::vector<int32_t>& boxes, const ssx_orb_params& prm, std::vector<ssx_keypoint>& kps) override
  {
    owner_ = std::this_thread::get_id();           // (the front-end calls come from the stream's thread)
    kps.assign((size_t)prm.nfeatures + 260 + 64, ssx_keypoint{});
    int32_t n = 0;
    const uint8_t* pinned = Pinned(img, 0);        // (the current left image: usually there already, from the frame's temporal LK)
    ssx_orb_detect_job& q = im_.det_req[k_];
    q = ssx_orb_detect_job{};
    q.img = pinned; q.stride = img.cols; q.boxes_xyxy = boxes.data(); q.n_boxes = (int32_t)(boxes.size() / 4);
    q.cap = (int32_t)kps.size(); q.kps_out = kps.data(); q.n_out = &n;
    im_.det_prm[k_] = prm; im_.lk_rows[k_] = img.rows; im_.lk_cols[k_] = img.cols;
    im_.SubmitAndWait(k_, St::PENDING_DET);
    if (n > (int32_t)kps.size()) {
      // (a grid returned more than the bound of ssx.h: once more with the size it asked for, on the stream's own context)
      LongOp op(im_, k_);
      kps.assign((size_t)n, ssx_keypoint{});
      frame_.check(ssx_orb_detect_boxes(frame_.get(), img.ptr(), img.cols, img.rows, img.cols, boxes.data(), (int32_t)(boxes.size() / 4), &prm, (int32_t)kps.size(),
                                        kps.data(), &n));
    }
    kps.resize(n);
  }

  void TrackLK(const Image& prev, const Image& next, const std::vector<float>& prev_pts, std::vector<float>& next_pts, std::vector<uint8_t>& status,
               bool temporal) override
  {
    owner_ = std::this_thread::get_id();           // (the front-end calls come from the stream's thread)
    const int n = (int)(prev_pts.size() / 2);
    status.assign(n, 0);
    ssx_lk_job& q = im_.lk_req[k_];
    q = ssx_lk_job{};
    q.slot = k_;
    q.n = n; q.prev_pts = prev_pts.data(); q.next_pts = next_pts.data(); q.status = status.data(); q.err = nullptr;
    im_.lk_rows[k_] = next.rows; im_.lk_cols[k_] = next.cols;
    if (!temporal) {
      // FindFeaturesInRight (frontend.cpp:346-428): left -> right of one frame, both pyramids built afresh, on the stereo LK context's
      // slot of this stream
      if (prev.rows != next.rows || prev.cols != next.cols) throw std::invalid_argument("TrackLK: image sizes differ");
      q.prev = Pinned(prev, 0); q.prev_stride = prev.cols;
      q.next = Pinned(next, 1, q.prev); q.next_stride = next.cols;
      im_.SubmitAndWait(k_, St::PENDING_LKS);
      return;
    }
    // the frame-to-frame chain: this stream's slot of the shared LK context keeps the pyramid of its last `next` image.  The new
    // image goes into the stream's pinned buffer (this thread copies it: S streams copy side by side) and is read from there by the
    // GPU -- no staging inside the batched call.
    const bool chained = prev.id != 0 && prev.id == chain_next_id_ && prev.rows == chain_rows_ && prev.cols == chain_cols_ && next.rows == prev.rows &&
                         next.cols == prev.cols;
    if (!chained) { q.prev = Pinned(prev, 1); q.prev_stride = prev.cols; }
    q.next = Pinned(next, 0, q.prev); q.next_stride = next.cols;
    chain_next_id_ = 0;                            // (a failed call leaves no chain)
    im_.SubmitAndWait(k_, St::PENDING_LK);
    chain_next_id_ = next.id; chain_rows_ = next.rows; chain_cols_ = next.cols;
  }

  int PoseOnly(double* pose_io, const double* K4, int M, const double* xyz, const double* uv, uint8_t* inlier) override
  {
    owner_ = std::this_thread::get_id();           // (the front-end calls come from the stream's thread)
    int32_t n_in = 0;
    ssx_pose_only_job& q = im_.po_req[k_];
    q = ssx_pose_only_job{};
    q.pose_io = pose_io; q.K4 = K4; q.M = M; q.xyz = xyz; q.uv = uv; q.rounds = 4; q.iters = 10; q.chi2_th = 5.991; q.huber_delta = 1.0;
    q.inlier_out = inlier; q.n_inliers = &n_in;
    im_.SubmitAndWait(k_, St::PENDING_PO);
    return n_in;
  }

  void Triangulate(int n, const double* uvL, const double* uvR, const ssx_stereo_rig& rig, const double* T_wc, double* xyz, uint8_t* ok) override
  {
    owner_ = std::this_thread::get_id();           // (the front-end calls come from the stream's thread)
    ssx_triangulate_job& q = im_.tri_req[k_];
    q = ssx_triangulate_job{};
    q.n = n; q.uvL = uvL; q.uvR = uvR; q.rig = &rig; q.T_wc = T_wc; q.xyz_out = xyz; q.ok_out = ok;
    im_.SubmitAndWait(k_, St::PENDING_TRI);
  }

  void BundleAdjust(const ssx_ba_problem& prob, const ssx_ba_options& opt, ssx_ba_result& res) override
  {
    // (Backend.Window: 0 -- the re-marshalled map, one window per call)
    if (std::this_thread::get_id() != owner_) {
      // the worker thread of an asynchronous backend: a context of its own, and the stream's slot is not touched (its thread is tracking)
      if (!async_ba_) async_ba_ = std::make_unique<ssx::Context>(im_.device);
      async_ba_->check(ssx_ba_solve(async_ba_->get(), &prob, &opt, &res));
      return;
    }
    LongOp op(im_, k_);
    frame_.check(ssx_ba_solve(frame_.get(), &prob, &opt, &res));
  }

  std::unique_ptr<BaWindow> MakeBaWindow(const double* K4, const double* cam_ext14, const ssx_ba_options& opt) override
  {
    return std::make_unique<BatchedBaWindow>(im_, k_, K4, cam_ext14, opt);
  }

 private:
  // the image in the stream's pinned buffer `which` (copied by this thread unless it is there already)
  // (keep: a buffer this call must not overwrite -- the other image of a two-image request)
  const uint8_t* Pinned(const Image& img, int which, const uint8_t* keep = nullptr)
  {
    const size_t bytes = (size_t)img.rows * img.cols;
    if (bytes > pin_bytes_) {
      if (own_pins_) { ssx_host_free(pin_[0]); ssx_host_free(pin_[1]); }
      pin_[0] = im_.PinSlice(k_, 0, bytes); pin_[1] = im_.PinSlice(k_, 1, bytes);
      own_pins_ = !pin_[0] || !pin_[1];
      if (own_pins_) { pin_[0] = static_cast<uint8_t*>(ssx_host_alloc(bytes)); pin_[1] = static_cast<uint8_t*>(ssx_host_alloc(bytes)); }
      if (!pin_[0] || !pin_[1]) throw std::runtime_error("StreamBatcher: no pinned memory for the stream's images");
      pin_bytes_ = bytes; pin_id_[0] = pin_id_[1] = 0;
    }
    if (img.id != 0 && pin_id_[which] == img.id) return pin_[which];
    if (img.id != 0 && pin_id_[which ^ 1] == img.id) return pin_[which ^ 1];
    if (keep && pin_[which] == keep) which ^= 1;
    std::memcpy(pin_[which], img.ptr(), bytes);
    pin_id_[which] = img.id;
    return pin_[which];
  }

  StreamBatcher::Impl& im_;
  int k_;
  std::thread::id owner_ = std::this_thread::get_id();
  std::unique_ptr<ssx::Context> async_ba_;
  ssx::Context frame_;
  uint8_t* pin_[2] = {nullptr, nullptr};
  uint64_t pin_id_[2] = {0, 0};
  size_t pin_bytes_ = 0;
  bool own_pins_ = false;
  uint64_t chain_next_id_ = 0;
  int chain_rows_ = 0, chain_cols_ = 0;
};

}  // namespace

// `cohorts` independent batchers, stream k in cohort k mod cohorts: while one cohort's batch is on the GPU the other cohort's streams do
// their host work (bookkeeping, the copy of the next frame into pinned memory, the wake-ups) -- two cohorts ping-pong
StreamBatcher::StreamBatcher(int device, int streams, int cohorts)
{
  streams = std::max(streams, 1);
  const int C = std::max(1, std::min(cohorts, streams));
  for (int c = 0; c < C; ++c) impls_.push_back(std::make_unique<Impl>(device, (streams - c + C - 1) / C));
  n_streams_ = streams;
}
StreamBatcher::~StreamBatcher() = default;

std::unique_ptr<Compute> StreamBatcher::MakeCompute(int k)
{
  if (k < 0 || k >= n_streams_) throw std::invalid_argument("StreamBatcher::MakeCompute: stream index out of range");
  const int C = (int)impls_.size();
  return std::make_unique<BatchedCompute>(*impls_[k % C], k / C);
}

void StreamBatcher::Finish(int k)
{
  const int C = (int)impls_.size();
  if (k >= 0 && k < n_streams_) impls_[k % C]->SetState(k / C, St::DONE);
}

void StreamBatcher::ResetStats()
{
  for (auto& im : impls_) {
    std::lock_guard<std::mutex> lk(im->m);
    im->st = Stats{};
  }
}

StreamBatcher::Stats StreamBatcher::stats()
{
  Stats t;
  for (auto& im : impls_) {
    std::lock_guard<std::mutex> lk(im->m);
    const Stats& a = im->st;
    t.lk_calls += a.lk_calls; t.lk_jobs += a.lk_jobs; t.po_calls += a.po_calls; t.po_jobs += a.po_jobs; t.ba_calls += a.ba_calls; t.ba_jobs += a.ba_jobs;
    t.kf_calls += a.kf_calls; t.kf_jobs += a.kf_jobs; t.lk_s += a.lk_s; t.po_s += a.po_s; t.ba_s += a.ba_s; t.kf_s += a.kf_s; t.wait_s += a.wait_s;
  }
  return t;
}

}  // namespace ssx::host
