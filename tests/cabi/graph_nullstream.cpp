// Torch-free reproducer / regression test for the fault DESIGN.md 8 records: "a hipGraph replay followed by further kernel launches on
// the LEGACY NULL STREAM without a host synchronisation ended in a GPU memory fault" (ROCm 7.2, 256 streams, ~80 ticks into the run).
// Built by hipcc on the GPU box (tests/test_gpu_stream.py), HIP runtime + the C ABI only.
//
//   part A  no kernel of this library: a captured graph {memset, kernel} replayed on the null stream, interleaved with direct
//           null-stream launches that read what the graph wrote, no host synchronisation for `ticks` rounds.  Every kernel checks the
//           value its predecessor must have left (a chain of non-commuting updates): if the runtime lets a null-stream launch overtake a
//           graph replayed on the null stream (or the reverse), the chain breaks and the count of broken links is printed.
//   part B  this library through its public ABI: B problems, a captured solve graph replayed with bmpc_graph_launch(g, NULL) and direct
//           bmpc_solve_batch launches on the null stream writing OTHER output buffers, `ticks` rounds without a host synchronisation,
//           then both outputs are compared bit for bit with a fully synchronised run.
// Output: one line "A broken=<n> B mismatch=<n> status=<ok|...>"; exit code 0 iff both are zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/boundmpc_hip.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

// v[i] <- 3 v[i] + 1 if it holds the expected predecessor value, else the link is counted as broken and the chain is re-seeded
__global__ void link_kernel(unsigned long long *v, int n, unsigned long long expect, unsigned long long *broken) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (v[i] != expect) atomicAdd(broken, 1ULL);
    v[i] = expect * 3ULL + 1ULL;
}

static int part_a(int ticks, unsigned long long *broken_out) {
    const int n = 1 << 16;
    unsigned long long *v = nullptr, *broken = nullptr, *zero_me = nullptr;
    CHK(hipMalloc(&v, n * sizeof(*v))); CHK(hipMalloc(&broken, sizeof(*broken))); CHK(hipMalloc(&zero_me, 64));
    CHK(hipMemset(v, 0, n * sizeof(*v))); CHK(hipMemset(broken, 0, sizeof(*broken)));
    // expected values are data-independent: e_0 = 0, e_{j+1} = 3 e_j + 1 (mod 2^64); the graph kernel takes its `expect` from the
    // capture, so one graph per tick would be needed -- instead the graph holds a kernel that only works on a SECOND array with a
    // fixed expectation cycle of length 1 (v2 <- 0 by memset, then link(0)), and the direct launches run the growing chain on v:
    // the cross-check is that every direct launch also reads v2 and requires the graph's result 1 there.
    unsigned long long *v2 = nullptr;
    CHK(hipMalloc(&v2, n * sizeof(*v2)));
    hipStream_t cs; CHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph; hipGraphExec_t exec;
    CHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    CHK(hipMemsetAsync(v2, 0, n * sizeof(*v2), cs));
    hipLaunchKernelGGL(link_kernel, dim3(n / 256), dim3(256), 0, cs, v2, n, 0ULL, broken);
    CHK(hipStreamEndCapture(cs, &graph));
    CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    unsigned long long e = 0;
    for (int t = 0; t < ticks; t++) {
        CHK(hipGraphLaunch(exec, nullptr));                                                       // graph on the legacy null stream
        hipLaunchKernelGGL(link_kernel, dim3(n / 256), dim3(256), 0, nullptr, v2, n, 1ULL, broken); // must see the graph's result (1), leaves 4
        hipLaunchKernelGGL(link_kernel, dim3(n / 256), dim3(256), 0, nullptr, v, n, e, broken);    // the growing chain
        e = e * 3ULL + 1ULL;
    }
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(broken_out, broken, sizeof(*broken), hipMemcpyDeviceToHost));
    hipGraphExecDestroy(exec); hipGraphDestroy(graph); hipStreamDestroy(cs);
    hipFree(v); hipFree(v2); hipFree(broken); hipFree(zero_me);
    return 0;
}

static int part_b(const char *file, int B, int ticks, unsigned long long *mismatch_out) {
    enum { NP = 505, NW = 440 };
    std::vector<double> hp((size_t)B * NP), hx0((size_t)B * NW);
    FILE *fh = fopen(file, "rb");
    if (!fh || fread(hp.data(), sizeof(double), hp.size(), fh) != hp.size() || fread(hx0.data(), sizeof(double), hx0.size(), fh) != hx0.size()) {
        fprintf(stderr, "cannot read %d problems from %s\n", B, file); return 3;
    }
    fclose(fh);
    bmpc_handle *h = nullptr; bmpc_graph *g = nullptr;
    if (bmpc_create(10, 4, 0.1, nullptr, &h) != BMPC_OK) return 3;
    double *p, *x0, *xg, *xd, *xref;
    CHK(hipMalloc(&p, hp.size() * 8)); CHK(hipMalloc(&x0, hx0.size() * 8));
    CHK(hipMalloc(&xg, hx0.size() * 8)); CHK(hipMalloc(&xd, hx0.size() * 8)); CHK(hipMalloc(&xref, hx0.size() * 8));
    CHK(hipMemcpy(p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(x0, hx0.data(), hx0.size() * 8, hipMemcpyHostToDevice));
    // reference: one synchronised solve
    if (bmpc_solve_batch(h, B, p, x0, xref, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != BMPC_OK) return 3;
    CHK(hipDeviceSynchronize());
    if (bmpc_graph_create(h, B, p, x0, nullptr, 0, xg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &g) != BMPC_OK) return 3;
    for (int t = 0; t < ticks; t++) {            // replays and direct launches mixed on the null stream, no host synchronisation
        if (bmpc_graph_launch(g, nullptr) != BMPC_OK) return 3;
        if (bmpc_solve_batch(h, B, p, x0, xd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != BMPC_OK) return 3;
    }
    CHK(hipDeviceSynchronize());
    std::vector<double> a(hx0.size()), b(hx0.size()), r(hx0.size());
    CHK(hipMemcpy(a.data(), xg, a.size() * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), xd, b.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r.data(), xref, r.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long mm = 0;
    for (size_t i = 0; i < r.size(); i++) mm += (memcmp(&a[i], &r[i], 8) != 0) + (memcmp(&b[i], &r[i], 8) != 0);
    *mismatch_out = mm;
    // a graph outlives bmpc_destroy of its handle (the handle's memory goes with the last graph) and refuses further launches
    if (bmpc_destroy(h) != BMPC_OK) return 3;
    if (bmpc_graph_launch(g, nullptr) != BMPC_ERR_ARG) { fprintf(stderr, "a graph of a destroyed handle must refuse to launch\n"); return 3; }
    if (bmpc_graph_destroy(g) != BMPC_OK) return 3;
    hipFree(p); hipFree(x0); hipFree(xg); hipFree(xd); hipFree(xref);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s problems.bin B ticks\n", argv[0]); return 2; }
    const int B = atoi(argv[2]), ticks = atoi(argv[3]);
    unsigned long long broken = 0, mismatch = 0;
    int rc = part_a(ticks, &broken);
    if (rc == 0) rc = part_b(argv[1], B, ticks, &mismatch);
    printf("A broken=%llu B mismatch=%llu status=%s\n", broken, mismatch, rc == 0 ? "ok" : "error");
    return (rc == 0 && broken == 0 && mismatch == 0) ? 0 : 1;
}
