// Launch plans: a captured training step re-issued as a plain launch list.
//
// The step (forward + 5 losses + backward + gradient reduces, train_no_amp.py:181-239) is static: fixed patch size, device-side
// token selection, device-resident dropout counters.  cwf.trainer captures it ONCE with stream capture (which records every launch
// of the step -- this library's kernels and the handful of torch fills/copies -- with its arguments and its cross-stream
// dependencies) and hands the resulting hipGraph_t to cwf_plan_create.  The plan orders the graph's nodes topologically (ties in
// the order the host issued them), assigns them to a small set of streams along the chains the capture had (the main stream's
// data-gradient chain, the weight-gradient side stream), and cwf_plan_run re-issues them: one hipModuleLaunchKernel per kernel
// node, hipEventRecord / hipStreamWaitEvent for the cross-stream edges.  On ROCm 7.2 hipGraphLaunch costs the host ~44 us per
// node of this graph (20.6 ms per step, more than Python's eager enqueue); the plan costs a plain launch per node.
//
// Marker nodes (cwf_plan_marker, captured on the communication stream behind the streams it waits for) cut the list into
// segments: cwf_plan_run stops after a marker, the caller enqueues the data-parallel all-reduce of the finished gradient slice on
// that stream, and continues -- the collective overlaps the rest of backward exactly as in eager mode.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include <vector>
#include "../../include/cwf_hip.h"

namespace {

__global__ void plan_marker_kernel(int id) { (void)id; }

enum NodeKind { NK_KERNEL = 0, NK_MEMSET = 1, NK_EMPTY = 3, NK_MARKER = 4 };
const int MAX_STREAMS = 4;

struct PlanNode {
    int kind = NK_EMPTY;
    int stream = 0;             // index into the run's stream set; -1 for markers (the caller's communication stream)
    int marker_id = -1;
    hipKernelNodeParams kp;     // NK_KERNEL (kernelParams / extra point into the graph node: the graph must outlive the plan)
    hipFunction_t fn = nullptr; // resolved once (hipGetFuncBySymbol); nullptr = launch through the host symbol
    hipMemsetParams ms;         // NK_MEMSET
    std::vector<int> wait;      // plan positions (on other streams) this node waits for
    bool record = false;        // a later node on another stream waits for this one
    hipEvent_t ev = nullptr;
};

struct Plan {
    std::vector<PlanNode> nodes;          // in issue order
    std::vector<hipStream_t> owned;       // streams 1.. (stream 0 is the caller's)
    std::vector<hipEvent_t> join_ev;      // one per owned stream
    hipEvent_t start_ev = nullptr;
    int n_streams = 1, n_kernels = 0, n_markers = 0, n_events = 0;
    int last_on[MAX_STREAMS] = {-1, -1, -1, -1};
};

char g_detail[256] = "";       // why the last cwf_plan_create refused a graph (cwf_plan_last_error)

void destroy(Plan* P) {
    for (auto& N : P->nodes)
        if (N.ev) (void)hipEventDestroy(N.ev);
    for (auto e : P->join_ev) (void)hipEventDestroy(e);
    if (P->start_ev) (void)hipEventDestroy(P->start_ev);
    for (auto s : P->owned) (void)hipStreamDestroy(s);
    delete P;
}

}  // namespace

extern "C" int cwf_plan_marker(int id, void* stream) {
    hipLaunchKernelGGL(plan_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, id);
    return (int)hipGetLastError();
}

extern "C" int cwf_plan_create(void* graph_, void** out) {
    if (!graph_ || !out) return CWF_E_BADARG;
    hipGraph_t graph = (hipGraph_t)graph_;
    size_t n = 0, ne = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return CWF_E_BADARG;
    std::vector<hipGraphNode_t> gn(n);
    e = hipGraphGetNodes(graph, gn.data(), &n);
    if (e != hipSuccess) return (int)e;
    e = hipGraphGetEdges(graph, nullptr, nullptr, &ne);
    if (e != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) {
        e = hipGraphGetEdges(graph, ef.data(), et.data(), &ne);
        if (e != hipSuccess) return (int)e;
    }
    // node handle -> creation index (the order the host issued the work in)
    std::vector<std::pair<hipGraphNode_t, int>> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = {gn[i], (int)i};
    std::sort(idx.begin(), idx.end());
    auto find = [&](hipGraphNode_t h) -> int {
        auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
        return (it != idx.end() && it->first == h) ? it->second : -1;
    };
    std::vector<std::vector<int>> succ(n), pred(n);
    for (size_t k = 0; k < ne; ++k) {
        int a = find(ef[k]), b = find(et[k]);
        if (a < 0 || b < 0) return CWF_E_BADARG;
        succ[a].push_back(b);
        pred[b].push_back(a);
    }
    // topological order, smallest creation index first
    std::vector<int> indeg(n), order;
    order.reserve(n);
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < n; ++i) {
        indeg[i] = (int)pred[i].size();
        if (!indeg[i]) ready.push((int)i);
    }
    while (!ready.empty()) {
        int u = ready.top();
        ready.pop();
        order.push_back(u);
        for (int v : succ[u])
            if (--indeg[v] == 0) ready.push(v);
    }
    if (order.size() != n) return CWF_E_BADARG;
    std::vector<int> pos(n);
    for (size_t i = 0; i < n; ++i) pos[order[i]] = (int)i;

    Plan* P = new Plan();
    P->nodes.resize(n);
    void* marker_fn = (void*)plan_marker_kernel;
    for (size_t i = 0; i < n; ++i) {
        PlanNode& N = P->nodes[i];
        hipGraphNode_t h = gn[order[i]];
        hipGraphNodeType ty;
        e = hipGraphNodeGetType(h, &ty);
        if (e != hipSuccess) { destroy(P); return (int)e; }
        if (ty == hipGraphNodeTypeKernel) {
            e = hipGraphKernelNodeGetParams(h, &N.kp);
            if (e != hipSuccess) { destroy(P); return (int)e; }
            if (N.kp.func == marker_fn && N.kp.kernelParams) {
                N.kind = NK_MARKER;
                N.marker_id = *(int*)N.kp.kernelParams[0];
                P->n_markers++;
            } else {
                N.kind = NK_KERNEL;
                if (hipGetFuncBySymbol(&N.fn, N.kp.func) != hipSuccess) {
                    (void)hipGetLastError();
                    N.fn = nullptr;
                    if (!N.kp.kernelParams) {
                        snprintf(g_detail, sizeof g_detail, "kernel node %zu: no function handle and no parameter array", i);
                        destroy(P);
                        return CWF_E_TOOLARGE;
                    }
                }
                P->n_kernels++;
            }
        } else if (ty == hipGraphNodeTypeMemset) {
            e = hipGraphMemsetNodeGetParams(h, &N.ms);
            if (e != hipSuccess) { destroy(P); return (int)e; }
            if (N.ms.height > 1 || !(N.ms.elementSize == 1 || N.ms.elementSize == 2 || N.ms.elementSize == 4)) {
                snprintf(g_detail, sizeof g_detail, "memset node %zu: elementSize %u width %zu height %zu pitch %zu", i, N.ms.elementSize, N.ms.width, N.ms.height, N.ms.pitch);
                destroy(P);
                return CWF_E_TOOLARGE;
            }
            N.kind = NK_MEMSET;
        } else if (ty == hipGraphNodeTypeMemcpy) {
            // hipMemcpyAsync under capture makes a 1-D copy node, for which the runtime has no parameter getter (the 3-D getter
            // returns garbage on it): refuse copy nodes altogether rather than guess.  The step this library captures has none
            // (its copies are kernels); a graph that does is replayed by the caller with hipGraphLaunch.
            snprintf(g_detail, sizeof g_detail, "memcpy node at position %zu (copy nodes are not supported: no 1-D parameter getter)", i);
            destroy(P);
            return CWF_E_TOOLARGE;
        } else if (ty == hipGraphNodeTypeEmpty) {
            N.kind = NK_EMPTY;
        } else {
            snprintf(g_detail, sizeof g_detail, "node %zu of type %d (host / child-graph / external-event nodes are not supported)", i, (int)ty);
            destroy(P);
            return CWF_E_TOOLARGE;       // not a step this library captured
        }
    }
    // ---- chains.  depth[v] = longest path from v to a sink.  A node hands its stream to ONE successor: a successor that has no other
    // predecessor if there is one (under stream capture the next node of the SAME stream depends on its predecessor alone, whereas a
    // node of another stream that waited for an event here also depends on that stream's tail), else -- and among several such -- the
    // one with the largest depth (the main stream's data-gradient chain is by far the longest).  Depth alone is not enough: late in
    // backward the serialised weight-gradient chain is DEEPER than what is left of the main stream, and handing stream 0 to it split
    // the main chain over extra streams (4 streams, 33 ms per step instead of 19).  Every other successor continues another
    // predecessor's stream, takes a side stream whose chain has ended, or opens one.  Any assignment is CORRECT (same-stream edges
    // are FIFO order, cross-stream edges get an event); the choice only decides what may overlap.
    std::vector<int> depth(n, 1);
    for (int i = (int)n - 1; i >= 0; --i) {
        int u = order[i];
        for (int v : succ[u]) depth[u] = std::max(depth[u], depth[v] + 1);
    }
    // The weight-gradient chain is recognised by its kernels (weight-gradient kernels, their operand conversions and slab reduces):
    // those nodes ALWAYS go to stream 1, the low-priority stream, wherever the capture had them, and take no part in the chain
    // logic of the other nodes -- a structural guess that put them on a high-priority stream, or the main chain on the low-priority
    // one, cost 30-40 % of the step (any assignment is correct; the events below carry every cross-stream edge).
    std::vector<char> side(n, 0);
    bool any_side = false;
    for (size_t i = 0; i < n; ++i) {
        const PlanNode& N = P->nodes[i];
        if (N.kind != NK_KERNEL) continue;
        const char* nm = hipKernelNameRefByPtr(N.kp.func, nullptr);
        // (+ the bias-gradient sums of the transposed convs: in_reduce_kernel<0> + stats_channel_sum_kernel, cwf.kernels.channel_sum_to)
        if (nm && (strstr(nm, "wgrad") || strstr(nm, "stats_channel_sum_kernel") ||
                   strstr(nm, "in_reduce_kernelILi0") || strstr(nm, "in_reduce_kernel<0"))) { side[order[i]] = 1; any_side = true; }
        else if (nm && strstr(nm, "to_bf16_kernel")) side[order[i]] = 2;      // decided below: a conversion belongs where its readers are
    }
    (void)hipGetLastError();
    for (int i = (int)n - 1; i >= 0; --i) {               // (reverse topological order: a conversion feeding a conversion is handled too)
        const int u = order[i];
        if (side[u] != 2) continue;
        bool all_side = !succ[u].empty();
        for (int v : succ[u]) all_side = all_side && side[v] == 1;
        side[u] = all_side ? 1 : 0;
        any_side = any_side || all_side;
    }
    // Everything that is not part of the weight-gradient chain stays on the caller's stream, in the order the host issued it (a valid
    // topological order: ties above are broken by creation index).  The chain logic below (CWF_PLAN_CHAINS=1) can spread independent
    // branches over further streams, but on this runtime a third stream made every step 60 % SLOWER (three plan streams: 31 ms per
    // step against 18), and a chain that breaks at a weight-gradient node would open one.
    static const bool chains = getenv("CWF_PLAN_CHAINS") != nullptr;
    std::vector<int> heir(n, -1);
    for (size_t u = 0; u < n; ++u) {
        int best = -1;
        for (int v : succ[u]) {
            if (P->nodes[pos[v]].kind == NK_MARKER || side[v]) continue;
            if (best < 0) { best = v; continue; }
            const bool sv = pred[v].size() == 1, sb = pred[best].size() == 1;
            if (sv != sb) { if (sv) best = v; continue; }
            if (depth[v] > depth[best] || (depth[v] == depth[best] && pos[v] < pos[best])) best = v;
        }
        heir[u] = best;
    }
    int tail[MAX_STREAMS] = {-1, -1, -1, -1};                 // creation index of the last node on each stream
    int waited[MAX_STREAMS + 1][MAX_STREAMS];                 // [waiting stream (MAX_STREAMS = marker stream)][source] -> last position waited for
    for (auto& row : waited) for (int& w : row) w = -1;
    std::vector<int> stream_of(n, -2);
    int n_streams = any_side ? 2 : 1;                      // stream 1 is reserved for the weight-gradient chain
    bool any = false;
    for (size_t i = 0; i < n; ++i) {
        int v = order[i];
        PlanNode& N = P->nodes[i];
        if (N.kind == NK_MARKER) {
            N.stream = -1;
        } else if (side[v]) {
            N.stream = 1;
            tail[1] = v;
        } else if (!chains) {
            N.stream = 0;
            tail[0] = v;
            any = true;
        } else {
            int s = -1;
            for (int p : pred[v]) {
                int sp = stream_of[p];
                if (sp >= 0 && sp != 1 && tail[sp] == p && heir[p] == v && (s < 0 || sp < s)) s = sp;
            }
            if (s < 0 && !any) s = 0;
            for (int c = 2; s < 0 && c < n_streams; ++c) {
                int t = tail[c];
                if (t < 0 || heir[t] < 0 || pos[heir[t]] < (int)i) s = c;         // that chain has ended
            }
            if (s < 0 && n_streams < MAX_STREAMS) { if (n_streams < 2) n_streams = 2; s = n_streams++; }
            if (s < 0) s = 0;
            N.stream = s;
            tail[s] = v;
            any = true;
        }
        stream_of[v] = N.stream;
        std::vector<int> w;
        for (int p : pred[v]) {
            const PlanNode& Q = P->nodes[pos[p]];
            if (Q.kind == NK_MARKER) {                        // a marker orders nothing on the device beyond what IT waited for
                for (int x : Q.wait)
                    if (P->nodes[x].stream != N.stream) w.push_back(x);
            } else if (Q.stream != N.stream) {
                w.push_back(pos[p]);
            }
        }
        int me = N.stream < 0 ? MAX_STREAMS : N.stream;
        for (int c = 0; c < MAX_STREAMS; ++c) {               // waits are cumulative per source stream: the latest node is enough
            int last = -1;
            for (int x : w)
                if (P->nodes[x].stream == c) last = std::max(last, x);
            if (last > waited[me][c]) {
                N.wait.push_back(last);
                if (N.kind != NK_MARKER) waited[me][c] = last; // (the marker stream is the caller's: assume nothing across calls)
                P->nodes[last].record = true;
            }
        }
    }
    P->n_streams = n_streams;
    for (int c = 0; c < MAX_STREAMS; ++c) P->last_on[c] = tail[c] < 0 ? -1 : pos[tail[c]];
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int c = 1; c < n_streams; ++c) {
        hipStream_t st;
        hipEvent_t ev;
        // stream 1 = the weight-gradient chain: the runtime's LOW priority (its own hardware-queue pool and the right scheduling hint
        // beside the data-gradient chain, see cwf.kernels.HipBackend.wgrad_stream); further streams: high
        static const char* sp = getenv("CWF_PLAN_SIDE_PRIO");          // A/B: "normal" / "high" for the weight-gradient stream (default low)
        int side_prio = least;
        if (sp && !strcmp(sp, "normal")) side_prio = (least + greatest) / 2; else if (sp && !strcmp(sp, "high")) side_prio = greatest;
        e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, c == 1 ? side_prio : greatest);
        if (e != hipSuccess) { destroy(P); return (int)e; }
        P->owned.push_back(st);
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) { destroy(P); return (int)e; }
        P->join_ev.push_back(ev);
    }
    e = hipEventCreateWithFlags(&P->start_ev, hipEventDisableTiming);
    if (e != hipSuccess) { destroy(P); return (int)e; }
    for (auto& N : P->nodes)
        if (N.record) {
            e = hipEventCreateWithFlags(&N.ev, hipEventDisableTiming);
            if (e != hipSuccess) { destroy(P); return (int)e; }
            P->n_events++;
        }
    *out = P;
    return 0;
}

extern "C" const char* cwf_plan_last_error(void) { return g_detail; }

extern "C" int cwf_plan_info(void* plan, int* info8) {
    if (!plan || !info8) return CWF_E_BADARG;
    Plan* P = (Plan*)plan;
    int per[MAX_STREAMS] = {0, 0, 0, 0};
    for (auto& N : P->nodes)
        if (N.stream >= 0 && N.stream < MAX_STREAMS) per[N.stream]++;
    info8[0] = (int)P->nodes.size(); info8[1] = P->n_kernels; info8[2] = P->n_markers; info8[3] = P->n_streams;
    info8[4] = P->n_events; info8[5] = per[0]; info8[6] = per[1]; info8[7] = per[2] + per[3];
    return 0;
}

// Issues nodes [start, ...) until the list ends or a marker has been processed; sets *next (= the node count when the list is
// finished) and *marker_id (-1 when finished).  main_stream carries chain 0; comm_stream (may be NULL for a plan without markers)
// receives the markers' waits: work the caller enqueues on it after the call is ordered behind everything the marker depended on.
// When the list finishes, main_stream has been made to wait for the side streams.
extern "C" int cwf_plan_run(void* plan, void* main_stream, void* comm_stream, int start, int* next, int* marker_id) {
    if (!plan || !next || !marker_id) return CWF_E_BADARG;
    Plan* P = (Plan*)plan;
    const int n = (int)P->nodes.size();
    if (start < 0 || start > n) return CWF_E_BADARG;
    hipStream_t st[MAX_STREAMS] = {(hipStream_t)main_stream, nullptr, nullptr, nullptr};
    for (int c = 1; c < P->n_streams; ++c) st[c] = P->owned[c - 1];
    hipError_t e;
    if (start == 0 && P->n_streams > 1) {
        // the side streams start behind whatever the caller enqueued on the main stream before the step (input copies, optimizer)
        e = hipEventRecord(P->start_ev, st[0]);
        if (e != hipSuccess) return (int)e;
        for (int c = 1; c < P->n_streams; ++c) {
            e = hipStreamWaitEvent(st[c], P->start_ev, 0);
            if (e != hipSuccess) return (int)e;
        }
    }
    for (int i = start; i < n; ++i) {
        PlanNode& N = P->nodes[i];
        if (N.kind == NK_MARKER && !comm_stream) return CWF_E_BADARG;
        hipStream_t s = N.kind == NK_MARKER ? (hipStream_t)comm_stream : st[N.stream];
        for (int w : N.wait) {
            e = hipStreamWaitEvent(s, P->nodes[w].ev, 0);
            if (e != hipSuccess) return (int)e;
        }
        e = hipSuccess;
        switch (N.kind) {
        case NK_KERNEL:
            if (N.fn)
                e = hipModuleLaunchKernel(N.fn, N.kp.gridDim.x, N.kp.gridDim.y, N.kp.gridDim.z, N.kp.blockDim.x, N.kp.blockDim.y,
                                          N.kp.blockDim.z, N.kp.sharedMemBytes, s, N.kp.kernelParams, N.kp.kernelParams ? nullptr : N.kp.extra);
            else
                e = hipLaunchKernel(N.kp.func, N.kp.gridDim, N.kp.blockDim, N.kp.kernelParams, N.kp.sharedMemBytes, s);
            break;
        case NK_MEMSET:
            if (N.ms.elementSize == 1) e = hipMemsetAsync(N.ms.dst, (int)N.ms.value, N.ms.width, s);
            else if (N.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)N.ms.dst, (unsigned short)N.ms.value, N.ms.width, s);
            else e = hipMemsetD32Async((hipDeviceptr_t)N.ms.dst, (int)N.ms.value, N.ms.width, s);
            break;
        default:
            break;
        }
        if (e != hipSuccess) return (int)e;
        if (N.record) {
            e = hipEventRecord(N.ev, s);
            if (e != hipSuccess) return (int)e;
        }
        if (N.kind == NK_MARKER) {
            *next = i + 1;
            *marker_id = N.marker_id;
            return 0;
        }
    }
    for (int c = 1; c < P->n_streams; ++c) {
        e = hipEventRecord(P->join_ev[c - 1], st[c]);
        if (e != hipSuccess) return (int)e;
        e = hipStreamWaitEvent(st[0], P->join_ev[c - 1], 0);
        if (e != hipSuccess) return (int)e;
    }
    *next = n;
    *marker_id = -1;
    return 0;
}

extern "C" int cwf_plan_destroy(void* plan) {
    if (plan) destroy((Plan*)plan);
    return 0;
}
