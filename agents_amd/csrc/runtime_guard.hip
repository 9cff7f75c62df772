// Guard against a HIP-runtime fault in hipGraphLaunch (ROCm 7.0, runtime version 70051831: the
// libamdhip64 this image's torch 2.10.0+rocm7.0 bundles).
//
// What happens there (disassembly of hip::GraphExec::Init / CreateStreams / Graph::UpdateStreams,
// tools/hip_queue_probe.py, profiles/r06_f_hipgraph_segv.txt): a graph with max_streams_ = n > 1
// creates n "parallel" streams when it is instantiated.  Every new HIP stream is put on the
// hardware queue (GPU_MAX_HW_QUEUES = 4) that currently has the fewest streams, so two of an
// exec's streams share a queue whenever the process's queues are unevenly loaded (a long-lived
// process that has destroyed other graph execs).  At launch UpdateStreams fills the n - 1 side
// slots from those n streams, SKIPPING each one whose queue is the launch stream's -- with no
// bounds check: two parallel streams on the launch stream's queue and it reads past the vector
// and dereferences what it finds (SIGSEGV; the 671st test of the GPU suite in rounds 5 and 6).
//
// An exec whose parallel streams all sit on DIFFERENT queues can meet at most one such skip, for
// any launch stream.  utils/graph.py therefore keeps the captured hipGraph and has THIS file
// instantiate it (aa_hip_graph_instantiate): where two of the fresh exec's parallel streams share
// a queue the exec is destroyed, one ballast stream is created (it takes the queue with the fewest
// streams, i.e. evens the loads out; with the exec gone -- an exec that is still alive keeps its
// own doubled streams in the very holes the next one would fall into) and the graph is
// instantiated again, until the streams are spread.  The loads pass through the all-equal state
// after at most (sum of the gaps) ballast streams, where n <= 4 new streams land on n different
// queues.  Capture-time only; replays are one hipGraphLaunch (aa_hip_graph_launch).
//
// This reads three fields of runtime-internal objects.  Offsets are those of runtime 70051831 and
// the entry points refuse (AA_ERR_UNSUPPORTED) on any other version, on which the guard is simply
// not applied (the fault has only been analysed, and only matters, on this build).
#include "common.h"
#include "agents_amd.h"
#include <vector>

namespace {
constexpr int kRuntime = 70051831;
constexpr size_t kExecMaxStreams = 0x48;   // hip::GraphExec: int max_streams_
constexpr size_t kExecParallelBegin = 0x1b8;   // std::vector<hip::Stream*> parallel_streams_
constexpr size_t kExecParallelEnd = 0x1c0;
constexpr size_t kStreamQueueObj = 0x1a8;  // hip::Stream: the object UpdateStreams compares ...
constexpr int kQueueIdSlot = 2;            // ... through this virtual (its queue's identity)

bool aa_runtime_is_known() {
  static int ver = -1;
  if (ver < 0) {
    int v = 0;
    ver = hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
  }
  return ver == kRuntime;
}

std::vector<hipStream_t>& aa_ballast() {
  static std::vector<hipStream_t> v;
  return v;
}
}  // namespace

extern "C" {

int aa_hip_graph_exec_spread(void* graph_exec, int32_t* n_streams_out,
                             int32_t* max_on_one_queue_out) {
  if (graph_exec == nullptr || n_streams_out == nullptr || max_on_one_queue_out == nullptr)
    return AA_ERR_INVALID;
  if (!aa_runtime_is_known()) return AA_ERR_UNSUPPORTED;
  const char* e = reinterpret_cast<const char*>(graph_exec);
  const int n = *reinterpret_cast<const int*>(e + kExecMaxStreams);
  void* const* b = *reinterpret_cast<void* const* const*>(e + kExecParallelBegin);
  void* const* en = *reinterpret_cast<void* const* const*>(e + kExecParallelEnd);
  *n_streams_out = n;
  *max_on_one_queue_out = 0;
  if (n <= 1) return b == en ? AA_OK : AA_ERR_UNSUPPORTED;   // one stream: no parallel list
  if (n > 64 || b == nullptr || en - b != n) return AA_ERR_UNSUPPORTED;   // not the layout we know
  void* ids[64];
  for (int i = 0; i < n; ++i) {
    const char* s = reinterpret_cast<const char*>(b[i]);
    if (s == nullptr) return AA_ERR_UNSUPPORTED;
    void* obj = *reinterpret_cast<void* const*>(s + kStreamQueueObj);
    if (obj == nullptr) return AA_ERR_UNSUPPORTED;
    using Fn = void* (*)(void*);
    Fn fn = reinterpret_cast<Fn>((*reinterpret_cast<void** const*>(obj))[kQueueIdSlot]);
    ids[i] = fn(obj);
  }
  int worst = 1;
  for (int i = 0; i < n; ++i) {
    int same = 0;
    for (int j = 0; j < n; ++j) same += ids[j] == ids[i] ? 1 : 0;
    worst = same > worst ? same : worst;
  }
  *max_on_one_queue_out = worst;
  return AA_OK;
}

int aa_hip_graph_instantiate(void* graph, int32_t guard, void** exec_out, int32_t* n_streams_out,
                             int32_t* max_on_one_queue_out, int32_t* attempts_out) {
  if (graph == nullptr || exec_out == nullptr) return AA_ERR_INVALID;
  int32_t n = 0, worst = 0, attempts = 0;
  hipGraphExec_t ex = nullptr;
  int rc = AA_OK;
  for (;;) {
    ++attempts;
    if (hipGraphInstantiate(&ex, (hipGraph_t)graph, nullptr, nullptr, 0) != hipSuccess) {
      rc = AA_ERR_LAUNCH;
      ex = nullptr;
      break;
    }
    const int src = aa_hip_graph_exec_spread(ex, &n, &worst);
    if (src == AA_ERR_UNSUPPORTED) {      // another runtime: keep the exec as it is
      n = -1;
      worst = -1;
      break;
    }
    if (src != AA_OK) { rc = src; break; }
    if (!guard || worst <= 1) break;
    if (attempts >= 256) { rc = AA_ERR_RANGE; break; }
    (void)hipGraphExecDestroy(ex);        // its streams leave their queues ...
    ex = nullptr;
    hipStream_t s = nullptr;              // ... and one ballast stream takes the emptiest
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      rc = AA_ERR_LAUNCH;
      break;
    }
    aa_ballast().push_back(s);
  }
  for (hipStream_t s : aa_ballast()) (void)hipStreamDestroy(s);
  aa_ballast().clear();
  if (rc != AA_OK && ex != nullptr) {
    (void)hipGraphExecDestroy(ex);
    ex = nullptr;
  }
  *exec_out = ex;
  if (n_streams_out) *n_streams_out = n;
  if (max_on_one_queue_out) *max_on_one_queue_out = worst;
  if (attempts_out) *attempts_out = attempts;
  return rc;
}

int aa_hip_graph_launch(void* graph_exec, void* stream) {
  if (graph_exec == nullptr) return AA_ERR_INVALID;
  return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess
             ? AA_OK : AA_ERR_LAUNCH;
}

int aa_hip_graph_exec_destroy(void* graph_exec) {
  if (graph_exec == nullptr) return AA_OK;
  return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? AA_OK : AA_ERR_LAUNCH;
}

}  // extern "C"
