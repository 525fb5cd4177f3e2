// Library plumbing: error string, device info, HIP-graph capture of a launch sequence.
#include "common.hpp"

#include <string.h>

static thread_local char g_err[512] = "";

void cvmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[160] = "";

void cvmi_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

extern "C" const char* cvmi_last_kernel(void) {      // read-and-clear, so a call that dispatches no tagged kernel reads as ""
  static thread_local char out[sizeof(g_kernel)];
  memcpy(out, g_kernel, sizeof(out));
  g_kernel[0] = 0;
  return out;
}

extern "C" int cvmi_version(void) { return CVMI_VERSION; }
extern "C" const char* cvmi_last_error(void) { return g_err; }

extern "C" size_t cvmi_desc_size(int kind) {
  switch (kind) {
    case CVMI_DESC_CONV: return sizeof(cvmi_conv_desc);
    case CVMI_DESC_C3K2: return sizeof(cvmi_c3k2_desc);
    case CVMI_DESC_DWPW: return sizeof(cvmi_dwpw_desc);
    case CVMI_DESC_ATTN: return sizeof(cvmi_attn_desc);
    default: return 0;
  }
}

extern "C" int cvmi_device_info(int device, int* out4) {
  CVMI_CHECK(out4, "device_info: null pointer");
  hipDeviceProp_t prop;
  CVMI_HIP(hipGetDeviceProperties(&prop, device));
  out4[0] = prop.multiProcessorCount;
  out4[1] = prop.warpSize;
  out4[2] = (int)prop.maxSharedMemoryPerMultiProcessor;
  int arch = 0;
  const char* g = strstr(prop.gcnArchName, "gfx");
  if (g) arch = atoi(g + 3);
  out4[3] = arch;
  return 0;
}

extern "C" int cvmi_graph_begin(cvmi_stream_t stream) {
  CVMI_CHECK(stream, "graph_begin: capture needs a non-default stream");
  CVMI_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

extern "C" int cvmi_graph_end(cvmi_stream_t stream, void** graph_exec_out) {
  CVMI_CHECK(stream && graph_exec_out, "graph_end: null argument");
  hipGraph_t graph = nullptr;
  CVMI_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  CVMI_CHECK(graph, "graph_end: capture produced no graph (a captured call failed)");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) CVMI_FAIL("hipGraphInstantiate failed: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)exec;
  return 0;
}

extern "C" int cvmi_graph_launch(void* graph_exec, cvmi_stream_t stream) {
  CVMI_CHECK(graph_exec, "graph_launch: null graph");
  CVMI_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

extern "C" int cvmi_graph_destroy(void* graph_exec) {
  if (graph_exec) CVMI_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}
