// Library-wide entry points: version, last-error string.
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

extern "C" void npvp_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* npvp_last_error(void) { return g_err; }
extern "C" int npvp_version(void) { return 100; }

// kernels launched by this library in this process so far (every launch site is an NPVP_LAUNCH, common.h)
namespace npvp { long long g_launches = 0; }
extern "C" long long npvp_launch_count(void) { return __atomic_load_n(&npvp::g_launches, __ATOMIC_RELAXED); }

// A HIP stream of the LOWEST priority the device offers (PyTorch only hands out normal / high).  The gradient stream
// (npvp_amd.ops.WgradStream) is created with it so that, whenever CUs free up, the kernels of the critical
// forward/backward chain are dispatched before the queued weight-gradient workgroups.  *least / *greatest receive the
// device's priority range (nullable).  The stream lives until npvp_stream_destroy.
extern "C" void* npvp_stream_create_low_priority(int* least, int* greatest) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
  if (least) *least = lo;
  if (greatest) *greatest = hi;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess) {
    npvp_set_error("stream_create_low_priority: hipStreamCreateWithPriority failed");
    return nullptr;
  }
  return (void*)s;
}
extern "C" int npvp_stream_destroy(void* s) {
  return hipStreamDestroy((hipStream_t)s) == hipSuccess ? 0 : -3;
}

// ---- timing events a HIP-graph capture can carry (bench.py's live roofline probe).  A plain hipEventRecord on a capturing stream
// only orders work INSIDE the graph: after a replay the event holds no timestamp.  hipEventRecordWithFlags(hipEventRecordExternal)
// makes the record a node of the graph that stamps the event at every replay, so a pair bracketing a kernel reads that kernel's
// time in the LAST replay.  On a stream that is not capturing the record is an ordinary one.  (PyTorch-ROCm refuses external
// events - "External events are disallowed in rocm" - hence these four entry points.)
extern "C" void* npvp_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) { npvp_set_error("event_create: hipEventCreate failed"); return nullptr; }
  return (void*)e;
}
extern "C" int npvp_event_record(void* ev, void* stream) {
  NPVP_CHECK_ARG(ev, "event_record: null event");
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
  if (cs != hipStreamCaptureStatusActive) {
    const hipError_t r = hipEventRecord((hipEvent_t)ev, st);
    if (r != hipSuccess) { npvp_set_error(hipGetErrorString(r)); return NPVP_ERR_LAUNCH; }
    return NPVP_OK;
  }
  // capturing: an event-record NODE behind everything captured so far on this stream, and the stream's capture continues from it
  // (hipEventRecordWithFlags(..., hipEventRecordExternal) is the one-call form; ROCm 7.2 answers it with "invalid argument", so
  // the node is added by hand)
  hipStreamCaptureStatus st2 = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t r = hipStreamGetCaptureInfo_v2(st, &st2, &id, &graph, &deps, &ndeps);
  if (r != hipSuccess || !graph) { (void)hipGetLastError(); npvp_set_error("event_record: hipStreamGetCaptureInfo_v2 failed"); return NPVP_ERR_LAUNCH; }
  hipGraphNode_t node = nullptr;
  r = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)ev);
  if (r != hipSuccess) { (void)hipGetLastError(); npvp_set_error("event_record: hipGraphAddEventRecordNode failed"); return NPVP_ERR_LAUNCH; }
  r = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
  if (r != hipSuccess) { (void)hipGetLastError(); npvp_set_error("event_record: hipStreamUpdateCaptureDependencies failed"); return NPVP_ERR_LAUNCH; }
  return NPVP_OK;
}
// milliseconds between two recorded events, both complete (the caller has synchronised); < 0: not ready / not recorded
extern "C" float npvp_event_elapsed_ms(void* e0, void* e1) {
  float ms = -1.0f;
  if (!e0 || !e1 || hipEventElapsedTime(&ms, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) { (void)hipGetLastError(); return -1.0f; }
  return ms;
}
extern "C" int npvp_event_destroy(void* ev) { return ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess ? NPVP_OK : NPVP_ERR_ARG; }

// What a captured step consists of.  counts[16] by hipGraphNodeType (0 kernel, 1 memcpy, 2 memset, 3 host, 4 child graph, 5 empty,
// 6 wait-event, 7 event-record, ...); memset_bytes[0 .. max_memsets) = the sizes of the first memset nodes.  Returns the number
// of nodes, or a negative error.  Why it exists: memset nodes are what the ROCm 7.2 prepared-packet replay does not execute reliably
// (profiles/r06_graph_alloc_hazard.txt) - trainer.GraphedTrainStep counts them and refuses that replay mode when it finds any.
extern "C" long long npvp_graph_node_counts(void* graph, long long* counts, long long* memset_bytes, int max_memsets) {
  if (!graph || !counts) { npvp_set_error("graph_node_counts: null argument"); return NPVP_ERR_ARG; }
  size_t n = 0;
  if (hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) { npvp_set_error("graph_node_counts: hipGraphGetNodes failed"); return NPVP_ERR_LAUNCH; }
  for (int i = 0; i < 16; ++i) counts[i] = 0;
  if (n == 0) return 0;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) { npvp_set_error("graph_node_counts: out of host memory"); return NPVP_ERR_ARG; }
  if (hipGraphGetNodes((hipGraph_t)graph, nodes, &n) != hipSuccess) { free(nodes); npvp_set_error("graph_node_counts: hipGraphGetNodes failed"); return NPVP_ERR_LAUNCH; }
  int sets = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) continue;
    const int ti = (int)t;
    counts[ti >= 0 && ti < 15 ? ti : 15] += 1;
    if (t == hipGraphNodeTypeMemset && memset_bytes && sets < max_memsets) {
      hipMemsetParams mp;
      memset_bytes[sets++] = hipGraphMemsetNodeGetParams(nodes[i], &mp) == hipSuccess
                                 ? (long long)mp.elementSize * (long long)mp.width * (long long)(mp.height ? mp.height : 1) : -1;
    }
  }
  free(nodes);
  return (long long)n;
}

