// Launch plans: a phase's kernel-launch SEQUENCE kept inside the library (include/mintime_hip.h, "Launch plans").
//
// The host side of the reference's step (train.py:332-378) is Python issuing ~790 launches through ctypes: 24 ms of enqueue time
// per 47 ms step.  A plan records, once, every entry point a phase calls (EfficientNet forward, TimeSformer forward, ...) together
// with its argument VALUES -- device pointers of buffers the caller keeps alive and at fixed addresses, shapes, scalars, the stream
// -- and the cross-stream dependencies (mt_plan_fork); mt_plan_run re-issues the whole sequence from C: the same entry points, the
// same host-side variant selection, the same kernels, on the same two streams.  Nothing is captured by the HIP runtime (whole-step
// hipGraph replay was measured slower on ROCm 7.2: the forked weight-gradient branches lose their overlap); this is the launch
// loop moved below the FFI.
//
// Every exported `mt_xxx(..., void* stream)` is a generated thunk (gen_plan.py -> plan_thunks.inc) around the kernels' own entry
// point, compiled under the name mti_xxx (plan_rename.h): call through, and append a closure if this thread is recording.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <vector>

#include "../../include/mintime_hip.h"
#include "common.hpp"

namespace mt {

struct PlanOp {
  std::function<int()> fn;
  int tag;          // 0 = untagged; bit index of mt_plan_run's probe_mask otherwise
  double work;      // bytes or flops the caller attributes to the launch (mt_plan_tag)
  void* stream;     // the stream the op was recorded on (probe events go there)
};

struct ProbeHit {
  int tag;
  double work;
  hipEvent_t e0, e1;
};

struct Plan {
  std::vector<PlanOp> ops;
  std::vector<hipEvent_t> fork_events;      // one per recorded mt_plan_fork, re-recorded on every run
  std::vector<hipEvent_t> free_timing;      // pool of timing events for probes
  std::vector<ProbeHit> hits;
  int device = 0;
  bool recording = false;

  void add(void* stream, std::function<int()> fn);
};

static thread_local Plan* g_recording = nullptr;
static thread_local int g_next_tag = 0;
static thread_local double g_next_work = 0.0;

Plan* plan_recording() { return g_recording; }

void Plan::add(void* stream, std::function<int()> fn) {
  ops.push_back(PlanOp{std::move(fn), g_next_tag, g_next_work, stream});
  g_next_tag = 0;
  g_next_work = 0.0;
}

// events for mt_plan_fork outside a recording: a per-thread ring (a wait captures the record that preceded it, so an event may be
// re-recorded while an older wait on it is still pending)
static hipEvent_t ring_event() {
  static thread_local hipEvent_t ring[64] = {nullptr};
  static thread_local int ring_dev[64];
  static thread_local unsigned next = 0;
  const unsigned i = next++ & 63;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (ring[i] && ring_dev[i] != dev) {
    (void)hipEventDestroy(ring[i]);
    ring[i] = nullptr;
  }
  if (!ring[i]) {
    if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    ring_dev[i] = dev;
  }
  return ring[i];
}

static int fork_with(hipEvent_t ev, void* from, void* to) {
  hipError_t e = hipEventRecord(ev, (hipStream_t)from);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to, ev, 0);
  if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_plan_fork: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace mt

using mt::fail;
using mt::Plan;

#include "plan_thunks.inc"

extern "C" int mt_plan_create(mt_plan** out) {
  if (!out) return fail(mt::MT_ERR_ARG, "mt_plan_create: null pointer");
  Plan* p = new Plan();
  (void)hipGetDevice(&p->device);
  *out = reinterpret_cast<mt_plan*>(p);
  return 0;
}

extern "C" int mt_plan_destroy(mt_plan* plan) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) return 0;
  if (mt::g_recording == p) mt::g_recording = nullptr;
  for (hipEvent_t e : p->fork_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->free_timing) (void)hipEventDestroy(e);
  for (auto& h : p->hits) {
    (void)hipEventDestroy(h.e0);
    (void)hipEventDestroy(h.e1);
  }
  delete p;
  return 0;
}

extern "C" int mt_plan_record_begin(mt_plan* plan) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) return fail(mt::MT_ERR_ARG, "mt_plan_record_begin: null plan");
  if (mt::g_recording) return fail(mt::MT_ERR_ARG, "mt_plan_record_begin: this thread is already recording a plan");
  if (!p->ops.empty()) return fail(mt::MT_ERR_ARG, "mt_plan_record_begin: the plan already holds a recording");
  p->recording = true;
  mt::g_recording = p;
  mt::g_next_tag = 0;
  return 0;
}

extern "C" int mt_plan_record_end(mt_plan* plan) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p || mt::g_recording != p) return fail(mt::MT_ERR_ARG, "mt_plan_record_end: this thread is not recording that plan");
  p->recording = false;
  mt::g_recording = nullptr;
  mt::g_next_tag = 0;
  return 0;
}

extern "C" int mt_plan_size(const mt_plan* plan) {
  const Plan* p = reinterpret_cast<const Plan*>(plan);
  return p ? (int)p->ops.size() : 0;
}

extern "C" int mt_plan_tag(int tag, double work) {
  if (tag < 0 || tag > 31) return fail(mt::MT_ERR_ARG, "mt_plan_tag: tag must lie in [0, 31]");
  mt::g_next_tag = tag;
  mt::g_next_work = work;
  return 0;
}

extern "C" int mt_plan_fork(void* from_stream, void* to_stream) {
  if (Plan* p = mt::plan_recording()) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(mt::MT_ERR_LAUNCH, "mt_plan_fork: event");
    p->fork_events.push_back(ev);
    const int rc = mt::fork_with(ev, from_stream, to_stream);
    if (rc == 0) p->add(to_stream, [=]() { return mt::fork_with(ev, from_stream, to_stream); });
    return rc;
  }
  hipEvent_t ev = mt::ring_event();
  if (!ev) return fail(mt::MT_ERR_LAUNCH, "mt_plan_fork: event");
  return mt::fork_with(ev, from_stream, to_stream);
}

extern "C" int mt_plan_run(mt_plan* plan, unsigned probe_mask) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p) return fail(mt::MT_ERR_ARG, "mt_plan_run: null plan");
  if (p->recording) return fail(mt::MT_ERR_ARG, "mt_plan_run: the plan is still recording");
  if (mt::g_recording) return fail(mt::MT_ERR_ARG, "mt_plan_run: this thread is recording another plan");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != p->device) return fail(mt::MT_ERR_ARG, "mt_plan_run: recorded on device %d, current device is %d", p->device, dev);
  const size_t n = p->ops.size();
  for (size_t i = 0; i < n; ++i) {
    mt::PlanOp& op = p->ops[i];
    const bool probe = op.tag && ((probe_mask >> op.tag) & 1u);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (probe) {
      for (hipEvent_t* e : {&e0, &e1}) {
        if (!p->free_timing.empty()) {
          *e = p->free_timing.back();
          p->free_timing.pop_back();
        } else if (hipEventCreate(e) != hipSuccess) {
          return fail(mt::MT_ERR_LAUNCH, "mt_plan_run: probe event");
        }
      }
      (void)hipEventRecord(e0, (hipStream_t)op.stream);
    }
    const int rc = op.fn();
    if (probe) {
      (void)hipEventRecord(e1, (hipStream_t)op.stream);
      p->hits.push_back(mt::ProbeHit{op.tag, op.work, e0, e1});
    }
    if (rc != 0) return rc;          // mt_last_error holds the entry point's message
  }
  return 0;
}

extern "C" int mt_plan_probe_read(mt_plan* plan, int tag, int* launches, double* ms, double* work) {
  Plan* p = reinterpret_cast<Plan*>(plan);
  if (!p || !launches || !ms || !work) return fail(mt::MT_ERR_ARG, "mt_plan_probe_read: null pointer");
  int n = 0;
  double t = 0.0, w = 0.0;
  std::vector<mt::ProbeHit> rest;
  for (auto& h : p->hits) {
    if (h.tag != tag) {
      rest.push_back(h);
      continue;
    }
    float el = 0.f;
    hipError_t e = hipEventSynchronize(h.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&el, h.e0, h.e1);
    if (e != hipSuccess) return fail(mt::MT_ERR_LAUNCH, "mt_plan_probe_read: %s", hipGetErrorString(e));
    t += el;
    w += h.work;
    ++n;
    p->free_timing.push_back(h.e0);
    p->free_timing.push_back(h.e1);
  }
  p->hits.swap(rest);
  *launches = n;
  *ms = t;
  *work = w;
  return 0;
}

// plain stream operations as entry points, so that a recorded phase can contain them (a zero fill of a gradient buffer, a copy into
// a static input buffer); compiled under mti_* like every other entry point that enqueues work
extern "C" int mti_memset_async(void* p, int value, int64_t bytes, void* stream) {
  if (bytes <= 0) return 0;
  if (!p) return fail(mt::MT_ERR_ARG, "mt_memset_async: null pointer");
  const hipError_t e = hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) return fail(mt::MT_ERR_LAUNCH, "mt_memset_async: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int mti_copy_async(void* dst, const void* src, int64_t bytes, void* stream) {
  if (bytes <= 0) return 0;
  if (!dst || !src) return fail(mt::MT_ERR_ARG, "mt_copy_async: null pointer");
  const hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return fail(mt::MT_ERR_LAUNCH, "mt_copy_async: %s", hipGetErrorString(e));
  return 0;
}
