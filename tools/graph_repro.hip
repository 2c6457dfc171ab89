// Pure-HIP (no torch, no libfrtm) reproducer of the stream-capture pattern behind the two hipGraphLaunch segfaults of round 5
// (gpurun_out/r5m/tests.log, gpurun_out/r5e/gpu_suite.log: SegNetwork._forward_graphed <- Tracker.track_window, in tests that also replay trunk graphs).
//
// What the product does (model/seg_network.py:225-262, csrc/backbone.hip: frtm_backbone_forward_at, _hip.py: capture):
//   * every graph is captured on ONE capture stream C of the process;
//   * a TRUNK graph forks inside the capture to the trunk's lane stream(s) through an event the trunk owns (fork), and joins through another (done);
//   * a REFINER graph forks to one shared side stream R through events that live as long as the graph, and joins the same way;
//   * graphs are instantiated, replayed many times on the tracker's stream M, and die with their tracker -- together with the trunk that owned the
//     lane streams / events (rounds 1-4: the trunk CREATED its lane streams and DESTROYED them with itself; round 5: a process-wide pool, never destroyed).
//
//   hipcc --offload-arch=gfx950 -O2 tools/graph_repro.hip -o /tmp/graph_repro && /tmp/graph_repro <mode> <iterations>
//   mode 0  lane streams created per "trunk" and destroyed with it (rounds 1-4), events destroyed with it, graph execs destroyed before / after (alternating)
//   mode 1  lane streams from a pool that is never destroyed (round 5's mitigation), events still destroyed with the trunk
//   mode 2  pool + events never destroyed (nothing that took part in a capture is ever destroyed)
//   mode 3  like 0, and older graph execs (captured across streams that are gone) KEEP being replayed for a few generations
//   mode 4  pool streams; the tracker's graphs are destroyed WHILE THEIR LAST REPLAYS ARE STILL IN FLIGHT (no synchronise before hipGraphExecDestroy: what
//           Python does when a CUDAGraph object loses its last reference -- an LRU eviction, a cache clear, a dead tracker -- right after a replay)
//   mode 5  like 4 with per-trunk streams / events destroyed in flight as well
// A SIGSEGV handler reports the iteration and the phase; exit code 139 then.
#include <hip/hip_runtime.h>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static volatile int g_iter = -1;
static const char* volatile g_phase = "start";
static void on_segv(int) {
  char buf[160];
  int n = snprintf(buf, sizeof buf, "SEGFAULT iteration %d phase %s\n", g_iter, g_phase);
  (void)!write(2, buf, n);
  _exit(139);
}

__global__ void k_work(float* p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 0.999f + a;
}

struct Trunk {                       // owns (or borrows) lane streams and the fork / join events, like frtm_backbone
  hipStream_t lane[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
  bool owns_streams = false;
};
struct Graph { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; std::vector<hipEvent_t> ev; Trunk* trunk = nullptr; };

static hipStream_t g_pool[2] = {nullptr, nullptr};
static std::vector<hipEvent_t> g_event_cemetery;      // mode 2: events are parked here instead of destroyed

static Trunk* make_trunk(int mode) {
  Trunk* t = new Trunk();
  t->owns_streams = (mode == 0 || mode == 3 || mode == 5);
  for (int l = 0; l < 2; ++l) {
    if (t->owns_streams) CK(hipStreamCreateWithFlags(&t->lane[l], hipStreamNonBlocking));
    else { if (!g_pool[l]) CK(hipStreamCreateWithFlags(&g_pool[l], hipStreamNonBlocking)); t->lane[l] = g_pool[l]; }
    CK(hipEventCreateWithFlags(&t->done[l], hipEventDisableTiming));
  }
  CK(hipEventCreateWithFlags(&t->fork, hipEventDisableTiming));
  return t;
}
static void kill_trunk(Trunk* t, int mode) {
  // (hipStreamDestroy of a busy stream is legal: the runtime defers it)
  for (int l = 0; l < 2; ++l) {
    if (mode == 2) g_event_cemetery.push_back(t->done[l]); else CK(hipEventDestroy(t->done[l]));
    if (t->owns_streams) CK(hipStreamDestroy(t->lane[l]));
  }
  if (mode == 2) g_event_cemetery.push_back(t->fork); else CK(hipEventDestroy(t->fork));
  delete t;
}

// a trunk pass inside a capture on C: fork to both lanes, ~20 kernels per lane, join
static void trunk_pass(Trunk* t, hipStream_t C, float* buf, int n) {
  CK(hipEventRecord(t->fork, C));
  for (int l = 0; l < 2; ++l) {
    CK(hipStreamWaitEvent(t->lane[l], t->fork, 0));
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, t->lane[l], buf + (size_t)(l + 1) * n, n, 0.01f * k);
    CK(hipEventRecord(t->done[l], t->lane[l]));
  }
  for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, C, buf, n, 0.02f * k);
  for (int l = 0; l < 2; ++l) CK(hipStreamWaitEvent(C, t->done[l], 0));
}

static Graph capture_trunk(Trunk* t, hipStream_t C, float* buf, int n) {
  Graph G; G.trunk = t;
  CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
  trunk_pass(t, C, buf, n);
  CK(hipStreamEndCapture(C, &G.g));
  CK(hipGraphInstantiate(&G.x, G.g, nullptr, nullptr, 0));
  return G;
}

// the refiner: four "levels", the three deep ones on the shared side stream R (two parallel branches), events owned by the graph
static Graph capture_refiner(hipStream_t C, hipStream_t R, float* buf, int n, int levels) {
  Graph G;
  CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
  auto order = [&](hipStream_t waiter, hipStream_t waited) {
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, waited)); CK(hipStreamWaitEvent(waiter, e, 0)); G.ev.push_back(e);
  };
  order(R, C);
  for (int L = 0; L < levels; ++L) {
    hipStream_t s = (L + 1 < levels) ? R : C;
    for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, s, buf + (size_t)(3 + L) * n, n, 0.03f * k);
  }
  order(C, R);
  for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, C, buf + (size_t)3 * n, n, 0.05f);
  CK(hipStreamEndCapture(C, &G.g));
  CK(hipGraphInstantiate(&G.x, G.g, nullptr, nullptr, 0));
  return G;
}
static void kill_graph(Graph& G, int mode) {
  if (G.x) CK(hipGraphExecDestroy(G.x));
  if (G.g) CK(hipGraphDestroy(G.g));
  for (hipEvent_t e : G.ev) { if (mode == 2) g_event_cemetery.push_back(e); else CK(hipEventDestroy(e)); }
  G.x = nullptr; G.g = nullptr; G.ev.clear();
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 500;
  signal(SIGSEGV, on_segv);
  int rv = 0, dv = 0;
  CK(hipRuntimeGetVersion(&rv)); CK(hipDriverGetVersion(&dv));
  printf("graph_repro mode %d, %d iterations; HIP runtime %d, driver %d\n", mode, iters, rv, dv);
  const int n = 1 << 16;
  float* buf; CK(hipMalloc(&buf, (size_t)8 * n * sizeof(float))); CK(hipMemset(buf, 0, (size_t)8 * n * sizeof(float)));
  hipStream_t C, R, M;            // capture stream, shared refiner side stream, the "tracker's" stream
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&R, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
  std::vector<Graph> old_trunk_graphs, old_refiner_graphs;
  std::vector<Trunk*> old_trunks;
  for (int it = 0; it < iters; ++it) {
    g_iter = it;
    // ---- a "tracker" is born: trunk with graphs, refiner with graphs of two window shapes ----
    g_phase = "make trunk"; Trunk* t = make_trunk(mode);
    g_phase = "capture trunk"; Graph T = capture_trunk(t, C, buf, n);
    g_phase = "capture refiner"; Graph Ra = capture_refiner(C, R, buf, n, 4), Rb = capture_refiner(C, R, buf, n, 3);
    // ---- a sequence: trunk pass (eager lanes on the FIRST pass: fork from M, as run_sequence does before a shape has been captured), then replays ----
    g_phase = "eager trunk pass"; trunk_pass(t, M, buf, n);
    for (int rep = 0; rep < 4; ++rep) {
      g_phase = "replay trunk"; CK(hipGraphLaunch(T.x, M));
      g_phase = "replay refiner a"; CK(hipGraphLaunch(Ra.x, M));
      g_phase = "replay refiner b"; CK(hipGraphLaunch(Rb.x, M));
      if (mode == 3) for (Graph& G : old_refiner_graphs) { g_phase = "replay OLD refiner graph"; CK(hipGraphLaunch(G.x, M)); }
      if (mode == 3) for (Graph& G : old_trunk_graphs) { g_phase = "replay OLD trunk graph (its lane streams are gone)"; CK(hipGraphLaunch(G.x, M)); }
    }
    if (mode < 4 || (it % 16) == 15) { g_phase = "sync"; CK(hipStreamSynchronize(M)); }
    // ---- the tracker dies; which of its parts goes first depends on the garbage collector: alternate ----
    if (mode == 3) {
      old_trunk_graphs.push_back(T); old_refiner_graphs.push_back(Ra);
      g_phase = "kill refiner b"; kill_graph(Rb, mode);
      g_phase = "kill trunk (streams, events) while its graph lives on"; kill_trunk(t, mode);
      if (old_trunk_graphs.size() > 3) {
        g_phase = "kill old graphs";
        kill_graph(old_trunk_graphs.front(), mode); old_trunk_graphs.erase(old_trunk_graphs.begin());
        kill_graph(old_refiner_graphs.front(), mode); old_refiner_graphs.erase(old_refiner_graphs.begin());
      }
    } else if (it & 1) {
      g_phase = "kill graphs first"; kill_graph(T, mode); kill_graph(Ra, mode); kill_graph(Rb, mode);
      g_phase = "kill trunk after"; kill_trunk(t, mode);
    } else {
      g_phase = "kill trunk first"; kill_trunk(t, mode);
      g_phase = "kill graphs after"; kill_graph(Ra, mode); kill_graph(T, mode); kill_graph(Rb, mode);
    }
    if (it % 100 == 0) { printf("iteration %d ok\n", it); fflush(stdout); }
  }
  g_phase = "final sync"; CK(hipDeviceSynchronize());
  printf("DONE mode %d: %d iterations without a crash\n", mode, iters);
  return 0;
}
