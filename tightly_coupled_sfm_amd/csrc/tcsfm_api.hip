// tcsfm_api.hip -- C ABI (include/tcsfm.h) over the gfx950 kernels in kernels.h.
// Host logic only: argument checking, scratch management, kernel sequencing on one HIP stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <math.h>

#include "../../include/tcsfm.h"
#include "kernels.h"
#include "dense_kernel.h"
#include "joint_kernel.h"
#include "dense_ref_kernel.h"
#include "scale_kernel.h"
#include "posenet_kernel.h"

using namespace tc;

// ---- guard bands (TCSFM_DEBUG_GUARDS=1; tcsfm_debug_check_guards): every device allocation of this library gets 4 KB filled with a pattern
// in front of it and behind it; the check reads the bands back and reports the allocations whose bands were written to.  A kernel's
// out-of-bounds WRITE into neighbouring scratch corrupts results silently (ADVICE r04 #1 went unnoticed for a round: hipMalloc's granularity
// hid it) and GPU AddressSanitizer is not available on this pool: the GPU test suite runs with the bands on instead (tests/conftest.py).
// Off (the default) the two functions are hipMalloc / hipFree themselves.
#include <mutex>
namespace tcguard {
constexpr size_t kBand = 4096;
constexpr unsigned char kFill = 0xA5;
struct Rec { char *base; size_t bytes; int line; bool reported; };
inline std::mutex &mtx() { static std::mutex m; return m; }
inline std::vector<Rec> &recs() { static std::vector<Rec> r; return r; }
inline bool on() { static const bool v = getenv("TCSFM_DEBUG_GUARDS") && atoi(getenv("TCSFM_DEBUG_GUARDS")) != 0; return v; }
inline hipError_t alloc(void **p, size_t bytes, int line) {
    if (!on()) return (hipMalloc)(p, bytes);
    char *base = nullptr;
    const size_t padded = (bytes + 255) / 256 * 256;            // (the rear band starts at the next 256-byte boundary: the user pointer keeps hipMalloc's alignment)
    hipError_t e = (hipMalloc)((void **)&base, padded + 2 * kBand);
    if (e != hipSuccess) return e;
    e = hipMemset(base, kFill, kBand);
    if (e == hipSuccess) e = hipMemset(base + kBand + bytes, kFill, padded - bytes + kBand);
    if (e != hipSuccess) { (void)(hipFree)(base); return e; }
    std::lock_guard<std::mutex> lk(mtx());
    recs().push_back({base, bytes, line, false});
    *p = base + kBand;
    return hipSuccess;
}
// number of bytes of the two bands of r that no longer hold the pattern (device synchronised by the copies)
inline long damaged(const Rec &r, long *first_off) {
    const size_t padded = (r.bytes + 255) / 256 * 256, rear = padded - r.bytes + kBand;
    std::vector<unsigned char> host(kBand + rear);
    if (hipMemcpy(host.data(), r.base, kBand, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(host.data() + kBand, r.base + kBand + r.bytes, rear, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long bad = 0;
    for (size_t i = 0; i < host.size(); i++)
        if (host[i] != kFill) { if (!bad && first_off) *first_off = i < kBand ? (long)i - (long)kBand : (long)(r.bytes + (i - kBand)); bad++; }
    return bad;
}
inline hipError_t release(void *p) {
    if (!on() || !p) return (hipFree)(p);
    std::lock_guard<std::mutex> lk(mtx());
    for (size_t i = 0; i < recs().size(); i++)
        if (recs()[i].base + kBand == (char *)p) {
            long off = 0;
            const long bad = recs()[i].reported ? 0 : damaged(recs()[i], &off);
            if (bad > 0) fprintf(stderr, "tcsfm guard: allocation of %zu bytes (tcsfm_api.hip:%d) freed with %ld band bytes overwritten, first at offset %ld\n", recs()[i].bytes, recs()[i].line, bad, off);
            char *base = recs()[i].base;
            recs().erase(recs().begin() + i);
            return (hipFree)(base);
        }
    return (hipFree)(p);      // (not one of ours)
}
}  // namespace tcguard
#define hipMalloc(p, bytes) tcguard::alloc((void **)(p), (bytes), __LINE__)
#define hipFree(p) tcguard::release((void *)(p))

namespace {

// tile geometry of the hot kernel (one place to retune)
#ifndef TC_TILE_H
#define TC_TILE_H 16
#endif
constexpr int TILE_W = 32, TILE_H = TC_TILE_H, TILE_NT = TILE_W * TILE_H;      // (A/B builds: -DTC_TILE_H=8 -> 32 x 8 tiles of 256 threads)

thread_local std::string g_create_error;

struct HostStage {  // device staging for host-pointer calls
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct tcsfm_ctx {
    int device = 0, H = 0, W = 0, max_pairs = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    float4 *tgtpack = nullptr, *srcpack = nullptr;
    float *depth_work = nullptr;
    float *partials = nullptr, *blockrec = nullptr;
    int *tickets = nullptr;
    PairState *state = nullptr;
    PairConst *pconst = nullptr;
    double *lin_out = nullptr;   // device [max_pairs][7*7+7+4]
    float *pose_dev = nullptr, *ls_dev = nullptr, *K_dev = nullptr, *stats_dev = nullptr;
    int tiles_x = 0, tiles_y = 0, nblk = 0, ngrp = 0, ngrp_pad = 0, stats_cap_iters = 0;
    int nblk_alloc = 0, ngrp_alloc = 0;   // scratch capacity (covers the 32x8 tiling of the dense kernel too)
    float *dense_rec = nullptr, *depth0 = nullptr;   // dense mode scratch, allocated on first use
    float *dense_rec2 = nullptr, *depth_alt = nullptr;   // ... second record / depth buffers of the fused back-substitution (ping-pong)
    float *dense_rec_acc = nullptr, *depth_acc = nullptr;   // dense LM: accepted per-pixel records / depth maps
    int *lm_accept = nullptr;
    double *delta = nullptr;
    // joint dense mode (one depth map per target shared by its S forward pairs): per-pixel records, workgroup records, per-target state
    float *jrec = nullptr, *jrec_acc = nullptr, *jblockrec = nullptr, *jdepth_acc = nullptr;
    JointState *jstate = nullptr;
    double *jdelta = nullptr;
    double *jpart = nullptr; int *jtick = nullptr;     // split record sums of the joint solve (JointSolveParams::nsplit): [2 targets][8][NACC] fp64, tickets
    int jrec_S = 0;
    // dense mode on the reference's loss (dense_ref_kernel.h): mask counts, fixed-point scatter sums, linearisation export
    int *dref_norms = nullptr;
    long long *dref_ext = nullptr;
    double *dref_export = nullptr;
    hipStream_t aux_stream = nullptr;  // joint dense mode: the inverse pairs' refinement runs beside the forward group's (fork / join by events)
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    float *sel_maps = nullptr;   // dense window modes: the forward pairs' diff | valid maps, [2][max_pairs][H*W], allocated on first use
    unsigned *scale_keys = nullptr, *scale_hist = nullptr;   // scale recovery scratch (keys, 256 bins + 4 state words)
    long long *dbg_stamps = nullptr;  // TCSFM_DEBUG_STAMPS=1: 8 wall-clock stamps of the last k_solve launch (100 MHz ticks)
    std::vector<HostStage> stage;
    std::string err;
    // event profiling (tcsfm_profile_*): one (start, stop, class) triple per bracketed launch
    int *err_host = nullptr, *err_dev = nullptr;   // host-mapped status word of the device-side guards (host / device address)
    // lanes (tcsfm_set_lanes): lane k >= 1 is a child handle with its own stream and scratch, so that several refine calls are
    // in flight at once; lane 0 is this handle.  in_ev orders a lane behind the work queued on the parent's stream at call time,
    // done_ev is what tcsfm_lane_wait makes the parent's stream wait for.
    std::vector<tcsfm_ctx *> lanes;
    hipEvent_t in_ev = nullptr, done_ev = nullptr;
    std::vector<hipEvent_t> marks;   // tcsfm_lane_event: a ring of events handed out to the caller
    size_t mark_next = 0;
    // tcsfm_refine_sequence: device ring of frames, copy stream, per-slot / per-window events, pose staging (allocated on first use)
    float *seq_img = nullptr, *seq_depth = nullptr, *seq_pose_in = nullptr, *seq_pose_out = nullptr, *seq_ls_out = nullptr, *seq_K = nullptr;
    int seq_slots = 0, seq_K_n = 0;    // ring slots + mirror slots allocated; copies of K held by seq_K
    float4 *seq_fpack = nullptr;       // frame-level pack cache of the ring: [slots][H+2][W+2] (rgb, depth) + [slots][H][W] depth planes
    float *seq_fdepth = nullptr;
    int *pair_idx = nullptr;           // [2][max_pairs] ring slots of every pair's source pack / target depth plane (k_pack_cached)
    float *seq_dense = nullptr;        // tcsfm_refine_dense_sequence: refined depth maps of all windows, per-window order (on a lane: the
    size_t seq_dense_cap = 0;          // lane's stacked maps of one call)
    float *seq_dense_tmp = nullptr;    // ... the handle's own stacked maps of one call (lane 0)
    size_t seq_dense_tmp_cap = 0;
    size_t seq_pose_cap = 0;           // windows x pairs the pose staging holds
    hipStream_t seq_copy = nullptr;
    std::vector<hipEvent_t> seq_copied, seq_done, seq_raw;      // per chunk: frames usable by the lanes / per call: done / per chunk: raw frames landed
    hipStream_t seq_pack = nullptr;    // k_frame_pack runs here, behind the chunk's copy (event), beside the next chunk's copy
    struct KOk { const float *p; int n; };
    KOk K_ok[4] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}};   // device intrinsics pointers (and counts) that already passed the pinhole check
    int K_ok_next = 0;
    // tcsfm_set_graph_replay: repeated device-pointer refine calls (same arguments) replayed as ONE captured HIP graph
    struct CallKey { tcsfm_opts o; int N, win_B, win_S, pad; const void *p[10]; };
    struct CallGraph { CallKey key; hipGraphExec_t exec; int seen; unsigned long long used; };
    int graph_slots = 0;               // 0: off
    std::vector<CallGraph> graphs;
    unsigned long long graph_clock = 0;
    int graph_captures = 0, graph_replays = 0;
    bool capturing = false;
    // tcsfm_refine_window_queued: calls of one shape waiting to run as ONE launch sequence (tcsfm_set_coalesce / tcsfm_flush)
    struct PendingCall { const float *tgt, *srcs, *dt, *ds, *K, *pose_in; float *pose_out; float *depth_out; const float *ls_in; float *ls_out; };
                                       // depth_out: dense calls; ls_in / ls_out: pose + scale calls (or null)
    float *dref_smooth = nullptr;      // l_smooth: [targets][2] mean of the sigmoid disparity, the target's whole term (k_dref_smooth)
    bool lanes_serial = false;         // tcsfm_set_lanes' self-probe found that this process's streams do NOT run side by side (they slow each other
                                       // down: include/tcsfm.h "lanes"): the *_async calls run on the handle's own stream, one after the other
    float lane_probe[2] = {0.f, 0.f};  // the probe's figures: ms of the stand-in launches on ONE stream / alternating over TWO streams
    bool dref_dirty = true;            // the scatter sums (dref_ext, dref_ext_src) may be non-zero: a call that ran to its end leaves them zero
                                       // (k_dense_joint clears what it consumes); a fresh allocation, an export or a failed call does not
    // free source depth maps (opts.free_source_depths): the inverse pairs as groups of one source
    float *jrec_src = nullptr; JointState *jstate_src = nullptr; double *jdelta_src = nullptr; long long *dref_ext_src = nullptr;
    float *qres_rho_src = nullptr, *qres_rec_src = nullptr;      // ... in the quarter-resolution parametrisation
    double *pose_lin = nullptr;        // l_pose_consist: [2][max_pairs][12] transforms at the linearisation (k_solve, kernels.h)
    float *qres_rho = nullptr, *qres_rec = nullptr;      // TCSFM_DEPTH_QUARTER: [targets][H/4 * W/4] cell unknowns, [targets][cells][JREC] cell records
    int coal_max = 0;
    int pend_dense = 0;                // the waiting calls are dense-mode calls (tcsfm_refine_dense_window_queued)
    int coal_lanes = 1;                // merged sequences alternate over this many of the handle's streams (tcsfm_set_coalesce_lanes)
    unsigned coal_dirty = 0;           // bit l: lane l ran a merged sequence the handle's stream has not been ordered behind yet
    std::vector<PendingCall> pending;
    tcsfm_opts pend_opts;
    int pend_B = 0, pend_S = 0;
    int coal_batches = 0, coal_calls = 0;
    unsigned short *trace_bits = nullptr;   // tcsfm_debug_trace: caller-owned device buffers (null = off)
    int *trace_decide = nullptr;
    long long trace_bits_cap = 0, trace_decide_cap = 0;
    bool tickets_dirty = false;        // a failed call may have left group tickets non-zero
    bool profiling = false;
    unsigned long long *stamp_buf = nullptr;   // in-kernel workgroup stamps of the linearisation launches of a profile session
    size_t stamp_cap = 0, stamp_used = 0;      // capacity / use in (start, end) pairs
    std::vector<std::pair<size_t, size_t>> stamp_launch;   // (first pair, workgroups) of every stamped launch
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;
    size_t ev_used = 0;
};

namespace {

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (a process driving several
// GPUs from one thread keeps allocating on the device it selected).
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

#define HIPCHK(h, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            (h)->tickets_dirty = true;                                                               \
            return TCSFM_E_HIP;                                                                      \
        }                                                                                            \
    } while (0)

// Captured call graphs bake in the handle's scratch pointers and run on the stream they were last launched on: whoever frees or
// re-sizes such scratch, switches the stream or evicts an entry drains BOTH streams a replay can be on first.
void sync_graph_streams(tcsfm_ctx *c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->own_stream != c->stream) (void)hipStreamSynchronize(c->own_stream);
}
void drop_graphs(tcsfm_ctx *c) {
    if (c->graphs.empty()) return;
    sync_graph_streams(c);
    for (auto &g : c->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

int fail(tcsfm_ctx *h, int code, const char *msg) {
    h->err = msg;
    return code;
}

// The device-side pinhole guards (init_pair, k_ground) raise a host-mapped status word; it is reported -- as the documented
// TCSFM_E_INTRINSICS -- by the next call on the handle or by tcsfm_synchronize, without a device synchronisation of its own.
static bool k_known(const tcsfm_ctx *h, const float *K, int n) {
    for (const auto &k : h->K_ok) if (k.p == K && K && n <= k.n) return true;
    return false;
}
static void k_add(tcsfm_ctx *h, const float *K, int n) {
    for (auto &k : h->K_ok) if (k.p == K) { if (n > k.n) k.n = n; return; }
    h->K_ok[h->K_ok_next] = {K, n};
    h->K_ok_next = (h->K_ok_next + 1) % 4;
}
static void k_forget(tcsfm_ctx *h) { for (auto &k : h->K_ok) k = {nullptr, 0}; }

int pending_error(tcsfm_ctx *h) {
    if (h->err_host && *reinterpret_cast<volatile int *>(h->err_host)) {
        *reinterpret_cast<volatile int *>(h->err_host) = 0;
        k_forget(h);
        return fail(h, TCSFM_E_INTRINSICS, "an earlier asynchronous call was given non-pinhole intrinsics (detected on the device): its results are NaN");
    }
    return TCSFM_OK;
}

constexpr int kMaxAcc = AccLayout<7>::NACC;
constexpr int kLinOut = 7 * 7 + 7 + 4;

int check_common(tcsfm_ctx *h, const tcsfm_opts *o, int N) {
    if (!h) return TCSFM_E_ARG;
    if (!o) return fail(h, TCSFM_E_ARG, "opts is NULL");
    if (N < 1 || N > h->max_pairs) return fail(h, TCSFM_E_ARG, "N out of range for this handle (1..max_pairs)");
    if (o->refine != TCSFM_REFINE_POSE && o->refine != TCSFM_REFINE_POSE_SCALE) return fail(h, TCSFM_E_ARG, "opts.refine unsupported");
    if (o->solver != TCSFM_SOLVER_GN && o->solver != TCSFM_SOLVER_LM) return fail(h, TCSFM_E_ARG, "opts.solver unsupported");
    if (o->param != TCSFM_PARAM_SE3 && o->param != TCSFM_PARAM_EULER) return fail(h, TCSFM_E_ARG, "opts.param unsupported");
    if (o->n_iters < 0 || o->n_iters > 1000) return fail(h, TCSFM_E_ARG, "opts.n_iters out of range");
    if (o->depth_is_disp && !(o->min_depth > 0 && o->max_depth > o->min_depth)) return fail(h, TCSFM_E_ARG, "min_depth/max_depth invalid");
    if (o->window_rule != TCSFM_WINDOW_PAIR && o->window_rule != TCSFM_WINDOW_REFERENCE) return fail(h, TCSFM_E_ARG, "opts.window_rule unsupported");
    if (!(o->w_pose_consist >= 0.f) || (o->w_pose_consist > 0.f && (o->window_rule != TCSFM_WINDOW_REFERENCE || o->solver != TCSFM_SOLVER_GN ||
                                                                   o->refine != TCSFM_REFINE_POSE || o->param != TCSFM_PARAM_SE3)))
        return fail(h, TCSFM_E_ARG, "opts.w_pose_consist needs window_rule = TCSFM_WINDOW_REFERENCE, the Gauss-Newton solver, TCSFM_REFINE_POSE and the SE(3) chart");
    return TCSFM_OK;
}

// TCSFM_WINDOW_REFERENCE (window forms only): the coupling of the pairs' costs, see include/tcsfm.h
void apply_window_rule(const tcsfm_ctx *h, const tcsfm_opts *o, int win_B, int win_S, int N, LinParams &P, SolveParams &S) {
    if (!win_B || o->window_rule != TCSFM_WINDOW_REFERENCE) return;
    P.rule = 1;
    P.fwd_noauto = o->argmin ? 0 : win_B * win_S;      // optimizer.py:71-73: without argmin the forward term has no auto-mask
    S.rule = 1; S.grp_fwd = win_B * win_S; S.n_pairs = N;
    S.scale_fwd = o->argmin ? 1.0 : 0.25; S.scale_inv = 0.25;
    S.b_dc = (double)o->w_dc / ((double)win_B * win_S * (double)h->H * (double)h->W);   // :83-86: mean over all S*B maps
    if (o->w_pose_consist > 0.f) {       // :95-96: 0.1 (poses + poses_inv).abs().mean() over the S B x 6 entries
        S.w_pc = (double)o->w_pose_consist / (6.0 * win_B * win_S); S.pc_eps = (double)o->irls_eps; S.pose_lin = h->pose_lin;
    }
}

// Stage a host array on the device (slot-indexed scratch that grows on demand) or pass a device pointer through.
template <typename T>
int to_dev(tcsfm_ctx *h, const tcsfm_opts *o, int slot, const T *p, size_t count, const T **out) {
    if (!o->host_ptrs || !p) { *out = p; return TCSFM_OK; }
    if ((int)h->stage.size() <= slot) h->stage.resize(slot + 1);
    HostStage &s = h->stage[slot];
    size_t bytes = count * sizeof(T);
    if (s.cap < bytes) {
        if (s.p) HIPCHK(h, hipFree(s.p));
        s.p = nullptr; s.cap = 0;
        HIPCHK(h, hipMalloc(&s.p, bytes));
        s.cap = bytes;
    }
    HIPCHK(h, hipMemcpyAsync(s.p, p, bytes, hipMemcpyHostToDevice, h->stream));
    *out = (const T *)s.p;
    return TCSFM_OK;
}

// Output counterpart: returns a device buffer to write into; copy_back() moves it to the host pointer.
template <typename T>
int out_dev(tcsfm_ctx *h, const tcsfm_opts *o, int slot, T *p, size_t count, T **out) {
    if (!o->host_ptrs || !p) { *out = p; return TCSFM_OK; }
    if ((int)h->stage.size() <= slot) h->stage.resize(slot + 1);
    HostStage &s = h->stage[slot];
    size_t bytes = count * sizeof(T);
    if (s.cap < bytes) {
        if (s.p) HIPCHK(h, hipFree(s.p));
        s.p = nullptr; s.cap = 0;
        HIPCHK(h, hipMalloc(&s.p, bytes));
        s.cap = bytes;
    }
    *out = (T *)s.p;
    return TCSFM_OK;
}

template <typename T>
int copy_back(tcsfm_ctx *h, const tcsfm_opts *o, T *host, const T *dev, size_t count) {
    if (!o->host_ptrs || !host) return TCSFM_OK;
    HIPCHK(h, hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    return TCSFM_OK;
}

// intrinsics must be pinhole; checked on the host when they are host pointers, otherwise after a small D2H copy
int check_intrinsics(tcsfm_ctx *h, const tcsfm_opts *o, const float *K_host_or_dev, int n) {
    // Device intrinsics are validated with one blocking D2H copy the FIRST time a (pointer, count) is seen; repeated calls
    // on the same buffer (a sequence, the bench loop) stay fully asynchronous.  The device side guards independently:
    // init_pair() poisons the pose with NaN when K is not pinhole, so a buffer mutated behind our back still fails loudly.
    if (!o->host_ptrs && k_known(h, K_host_or_dev, n)) return TCSFM_OK;
    std::vector<float> k((size_t)n * 9);
    if (o->host_ptrs) memcpy(k.data(), K_host_or_dev, k.size() * sizeof(float));
    else {
        HIPCHK(h, hipMemcpyAsync(k.data(), K_host_or_dev, k.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    for (int i = 0; i < n; i++) {
        const float *K = &k[(size_t)i * 9];
        if (K[1] != 0.f || K[3] != 0.f || K[6] != 0.f || K[7] != 0.f || K[8] != 1.f || !(K[0] != 0.f) || !(K[4] != 0.f))
            return fail(h, TCSFM_E_INTRINSICS, "intrinsics must be pinhole [fx 0 cx; 0 fy cy; 0 0 1]");
    }
    if (!o->host_ptrs) k_add(h, K_host_or_dev, n);
    return TCSFM_OK;
}

// RAII bracket: records a start event now and a stop event at scope exit when profiling is on
struct ProfScope {
    tcsfm_ctx *h;
    bool on;
    ProfScope(tcsfm_ctx *h_, int cls) : h(h_), on(h_->profiling) {
        if (!on) return;
        if (h->ev_used + 2 > h->ev_pool.size()) {
            size_t old = h->ev_pool.size();
            h->ev_pool.resize(old + 64);
            for (size_t i = old; i < h->ev_pool.size(); i++)
                if (hipEventCreate(&h->ev_pool[i]) != hipSuccess) { h->ev_pool.resize(i); break; }   // no event, no bracket
            if (h->ev_used + 2 > h->ev_pool.size()) { on = false; return; }
        }
        h->ev_class.push_back(cls);
        (void)hipEventRecord(h->ev_pool[h->ev_used], h->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(h->ev_pool[h->ev_used + 1], h->stream);
        h->ev_used += 2;
    }
};

// while profiling, hand the next in-kernel stamp slots (one (start, end) pair per workgroup) to a linearisation launch
void take_stamp(tcsfm_ctx *h, LinParams &P, size_t workgroups) {
    P.stamp = nullptr;
    if (h->profiling && h->stamp_buf && h->stamp_used + workgroups <= h->stamp_cap) {
        P.stamp = h->stamp_buf + 2 * h->stamp_used;
        h->stamp_launch.emplace_back(h->stamp_used, workgroups);
        h->stamp_used += workgroups;
    }
}

// k_linearize's SSIM gradient in its adjoint form (kernels.h, ADJ) or as the round-3 neighbour-visiting pass B: TCSFM_ADJOINT=0/1,
// read once per process (default: see adjoint_default below; both forms are exact, they differ in rounding only)
constexpr int kAdjointDefault = 0;
bool use_adjoint() {
    static const int v = [] { const char *e = getenv("TCSFM_ADJOINT"); return e ? (atoi(e) != 0) : (kAdjointDefault != 0); }();
    return v != 0;
}

template <int NP, bool DC, int MODE, bool ADJ>
void launch_lin_a(tcsfm_ctx *h, const LinParams &P, int N) {
    dim3 grid(h->nblk, N), block(TILE_NT);
    const bool sel = MODE != MODE_MAPS && P.sel_S > 1;   // window form with the min over sources: selection inside the kernel
    if (MODE != MODE_MAPS && P.trace != nullptr) {       // parity tests: the decision-recording build
        if (sel) hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, true, true, ADJ>), grid, block, 0, h->stream, P);
        else hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, false, true, ADJ>), grid, block, 0, h->stream, P);
    } else if (MODE != MODE_MAPS && !ADJ && P.tshare != 0) {       // window forms of the pose modes: the shared-pack instantiations (kernels.h TSH)
        if (sel) hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, true, false, false, false, MODE != MODE_MAPS>), grid, block, 0, h->stream, P);
        else hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, false, false, false, false, MODE != MODE_MAPS>), grid, block, 0, h->stream, P);
    } else if (sel)
        hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, true, false, ADJ>), grid, block, 0, h->stream, P);
    else
        hipLaunchKernelGGL((k_linearize<NP, DC, MODE, TILE_W, TILE_H, TILE_NT, false, false, ADJ>), grid, block, 0, h->stream, P);
}
template <int NP, bool DC, int MODE>
void launch_lin_t(tcsfm_ctx *h, const LinParams &P, int N) {
    if (MODE == MODE_LIN && use_adjoint()) launch_lin_a<NP, DC, MODE, MODE == MODE_LIN>(h, P, N);
    else launch_lin_a<NP, DC, MODE, false>(h, P, N);
}

void launch_lin(tcsfm_ctx *h, const LinParams &P_in, int N, int np, bool dc, int mode, int prof_class = 0) {
    LinParams P = P_in;
    if (prof_class == 0) take_stamp(h, P, (size_t)h->nblk * N); else P.stamp = nullptr;
    P.one_generation = (size_t)h->nblk * N <= 512;          // 256 CUs x 2 resident workgroups
    ProfScope prof(h, prof_class);
    if (np == 6) {
        if (mode == MODE_MAPS) launch_lin_t<6, false, MODE_MAPS>(h, P, N);
        else if (mode == MODE_COST) launch_lin_t<6, false, MODE_COST>(h, P, N);
        else if (dc) launch_lin_t<6, true, MODE_LIN>(h, P, N);
        else launch_lin_t<6, false, MODE_LIN>(h, P, N);
    } else {
        if (mode == MODE_MAPS) launch_lin_t<7, false, MODE_MAPS>(h, P, N);
        else if (mode == MODE_COST) launch_lin_t<7, false, MODE_COST>(h, P, N);
        else if (dc) launch_lin_t<7, true, MODE_LIN>(h, P, N);
        else launch_lin_t<7, false, MODE_LIN>(h, P, N);
    }
}

// TCSFM_SOLVE_LEAN=1: the 1024-thread form of the solve kernel for SE(3) calls without the pose-consistency term -- measured SLOWER than
// the 256-thread kernel (6.0 vs 5.3 us in-kernel, profiles/r05_solve_ab.txt): default off, kept for A/B runs
bool use_lean_solve() {
    static const int v = [] { const char *e = getenv("TCSFM_SOLVE_LEAN"); return e ? atoi(e) : 0; }();
    return v != 0;
}
void launch_solve(tcsfm_ctx *h, const SolveParams &S, int N, int np) {
    ProfScope prof(h, 1);
    const bool lean = S.param == TCSFM_PARAM_SE3 && !(S.w_pc > 0.0) && use_lean_solve();      // (decided by the options of the call: every launch of a call takes the same kernel)
    if (np == 6) { if (lean) hipLaunchKernelGGL((k_solve_lean<6>), dim3(N), dim3(1024), 0, h->stream, S); else hipLaunchKernelGGL((k_solve<6>), dim3(N), dim3(TC_SOLVE_NT), 0, h->stream, S); }
    else { if (lean) hipLaunchKernelGGL((k_solve_lean<7>), dim3(N), dim3(1024), 0, h->stream, S); else hipLaunchKernelGGL((k_solve<7>), dim3(N), dim3(TC_SOLVE_NT), 0, h->stream, S); }
}

// decision trace (tcsfm_debug_trace) of linearisation `lin` of a call over N pairs: [lin][N][H*W] bits, [lin][N] decisions
void trace_at(const tcsfm_ctx *h, int lin, int N, LinParams &P, SolveParams &S) {
    const size_t hw = (size_t)h->H * h->W;
    P.trace = h->trace_bits ? h->trace_bits + (size_t)lin * N * hw : nullptr;
    S.trace_decide = h->trace_decide ? h->trace_decide + (size_t)lin * N : nullptr;
}
int trace_check(tcsfm_ctx *h, const tcsfm_opts *o, int N) {
    const long long n_lin = o->n_iters + (o->solver == TCSFM_SOLVER_LM && o->n_iters > 0 ? 1 : 0);
    if (h->trace_bits && n_lin * N * (long long)h->H * h->W > h->trace_bits_cap) return fail(h, TCSFM_E_ARG, "tcsfm_debug_trace: bits buffer too small for this call");
    if (h->trace_decide && n_lin * N > h->trace_decide_cap) return fail(h, TCSFM_E_ARG, "tcsfm_debug_trace: decide buffer too small for this call");
    return TCSFM_OK;
}

int np_of(const tcsfm_opts *o) { return o->refine == TCSFM_REFINE_POSE_SCALE ? 7 : 6; }
int nacc_of(int np) { return np == 6 ? AccLayout<6>::NACC : AccLayout<7>::NACC; }

InitParams init_params(tcsfm_ctx *h, const tcsfm_opts *o, int N, const float *pose, const float *ls, const float *K, int shared) {
    InitParams I;
    I.pose = pose; I.log_scale = ls; I.K = K; I.st = h->state; I.pc = h->pconst; I.N = N; I.shared_image = shared;
    I.lambda0 = o->lambda0;
    I.K_mod = 0;
    I.err = h->err_dev;
    I.pose_lin = (o->window_rule == TCSFM_WINDOW_REFERENCE && o->w_pose_consist > 0.f) ? h->pose_lin : nullptr;
    return I;
}

// init == nullptr: pack only.  Otherwise the pair initialisation rides in the same launch (needs N == Nimg).
int run_pack(tcsfm_ctx *h, const tcsfm_opts *o, int Nimg, const float *tgt, const float *src, const float *dt, const float *ds,
             const InitParams *init = nullptr, int win_B = 0, int win_S = 0, float *depth_copy = nullptr, const WinOff *wo = nullptr,
             const CoalTab *ct = nullptr, float *depth_copy3 = nullptr, int *zero_ints = nullptr, int zero_n = 0, float *const *ct_depth3 = nullptr,
             bool tshare = false) {
    PackParams P;
    P.depth_out2 = depth_copy; P.depth_out3 = depth_copy3; P.zero_ints = zero_ints; P.zero_n = zero_n;
    P.tshare = tshare ? 1 : 0;
    if (tshare && !(ct || win_B > 0)) return fail(h, TCSFM_E_ARG, "internal: the shared-pack form needs a window-form pack");
    P.c_out3 = (ct && ct_depth3) ? 1 : 0;
    for (int i = 0; i < TC_MAX_COAL; i++) P.c_depth_out3[i] = (ct && ct_depth3 && i < ct->ncall) ? ct_depth3[i] : nullptr;
    if (wo) P.win_off = *wo; else P.win_off.on = 0;
    memset(&P.init, 0, sizeof(P.init));
    P.win_B = win_B; P.win_S = win_S;
    if (init) {
        // group tickets must be zero when k_linearize starts: zeroed at create, re-zeroed by the reducers after every launch;
        // only a call that failed midway can leave them dirty
        if (h->tickets_dirty) {
            HIPCHK(h, hipMemsetAsync(h->tickets, 0, (size_t)h->max_pairs * h->ngrp_alloc * sizeof(int), h->stream));
            h->tickets_dirty = false;
        }
        P.init = *init;
    }
    P.tgt = tgt; P.src = src; P.depth_t = dt; P.depth_s = ds;
    P.tgtpack = h->tgtpack; P.srcpack = h->srcpack; P.depth_out = h->depth_work;
    P.H = h->H; P.W = h->W; P.N = Nimg;
    P.wl = o->w_l1 / 3.f; P.ws = o->w_ssim / 3.f;
    P.depth_is_disp = o->depth_is_disp;
    P.min_disp = o->depth_is_disp ? 1.f / o->max_depth : 0.f;
    P.max_disp = o->depth_is_disp ? 1.f / o->min_depth : 0.f;
    int hw = h->H * h->W;
    if (ct ? Nimg != 2 * ct->cS * ct->ncall * ct->cB : (win_B > 0 && Nimg != 2 * win_S * win_B))
        return fail(h, TCSFM_E_ARG, "internal: a window-form pack covers the 2 S B directed pairs of its windows");
    {
        ProfScope prof(h, 2);
        // window forms: one row of workgroups per FORWARD pair, which packs its inverse too (pack_body)
        const int rows = (ct || win_B > 0) ? Nimg / 2 : Nimg;
        const unsigned ptiles = (unsigned)(((h->W + PT_W - 1) / PT_W) * ((h->H + PT_H - 1) / PT_H));       // 64 x 4 pixel tiles (pack_body)
        if (ct) hipLaunchKernelGGL(k_pack_coal, dim3(ptiles, rows), dim3(256), 0, h->stream, P, *ct);
        else hipLaunchKernelGGL(k_pack, dim3(ptiles, rows), dim3(256), 0, h->stream, P);
    }
    HIPCHK(h, hipGetLastError());
    return TCSFM_OK;
}

int run_init(tcsfm_ctx *h, const tcsfm_opts *o, int N, const float *pose, const float *ls, const float *K, int shared) {
    // group tickets must be zero when k_linearize starts; the reducers re-zero them, this covers an aborted earlier call
    HIPCHK(h, hipMemsetAsync(h->tickets, 0, (size_t)h->max_pairs * h->ngrp_alloc * sizeof(int), h->stream));
    InitParams I = init_params(h, o, N, pose, ls, K, shared);
    hipLaunchKernelGGL(k_init, dim3((N + 63) / 64), dim3(64), 0, h->stream, I);
    HIPCHK(h, hipGetLastError());
    return TCSFM_OK;
}

// Small tile grids (KITTI 640x192: 240 workgroups per pair) skip the in-launch group reduction: the solve kernel sums the
// workgroup records itself with all loads in flight (one batch of <= 32 per thread), which is cheaper than the
// publish / ticket / last-arriver tail of every linearisation.  Larger grids keep the two-level reduction.
int direct_records(const tcsfm_ctx *h) { return h->nblk <= 256; }

LinParams lin_params(tcsfm_ctx *h, const tcsfm_opts *o, int np) {
    LinParams P;
    memset(&P, 0, sizeof(P));
    P.tgtpack = h->tgtpack; P.srcpack = h->srcpack; P.depth_t = h->depth_work; P.pc = h->pconst; P.partials = h->partials; P.blockrec = h->blockrec; P.tickets = h->tickets;
    P.H = h->H; P.W = h->W; P.tiles_x = h->tiles_x; P.tiles_y = h->tiles_y; P.nacc = nacc_of(np); P.ngrp = h->ngrp; P.ngrp_pad = h->ngrp_pad;
    P.wl = o->w_l1 / 3.f; P.ws = o->w_ssim / 3.f; P.eps = o->irls_eps; P.automask = o->automask;
    P.direct = direct_records(h);
    return P;
}

SolveParams solve_params(tcsfm_ctx *h, const tcsfm_opts *o, int np, int shared) {
    SolveParams S;
    memset(&S, 0, sizeof(S));
    S.partials = h->partials; S.st = h->state; S.pc = h->pconst; S.stats = nullptr; S.lin_out = h->lin_out;
    S.ngrp = h->ngrp; S.nacc = nacc_of(np); S.np = np; S.has_dc = o->w_dc > 0.f;
    if (direct_records(h)) { S.partials = h->blockrec; S.ngrp = h->nblk; }
    S.n_iters = o->n_iters; S.solver = o->solver; S.param = o->param;
    S.b_dc = (double)o->w_dc / ((double)h->H * (double)h->W);
    S.lambda_up = o->lambda_up; S.lambda_down = o->lambda_down; S.lambda_min = o->lambda_min;
    S.prior_scale = np == 7 ? (double)o->prior_scale : 0.0;
    S.shared_image = shared;
    S.dbg = h->dbg_stamps;
    return S;
}

}  // namespace

namespace {
// Scratch of the joint dense kernels, allocated ONCE on first use (a captured call graph -- tcsfm_set_graph_replay -- bakes these pointers
// in, so they are never freed or re-sized while the handle lives).  The per-PIXEL records (jrec, jrec_acc) and the workgroup records are
// sized in floats for the largest layout any call can need: S sources -> max_pairs / (2 S) targets x JREC(S) floats, largest at S = JMAXS
// (20 floats x max_pairs / 4 targets; S = 1 under the reference's loss needs 8 x max_pairs / 2).  The per-TARGET arrays (state, step,
// accepted depth) are sized for the most targets any call can have: (max_pairs + 1) / 2, reached at S = 1 (they were sized
// for S >= 2 and the S = 1 reference-loss mode indexed past them).
constexpr int kJointSplitMax = 8;
int joint_scratch(tcsfm_ctx *h) {
    if (h->jrec) return TCSFM_OK;
    using JM = JointLayout<JMAXS>;
    const size_t hw = (size_t)h->H * h->W, n = h->max_pairs;
    const size_t nb = (n + 3) / 4, nt = (n + 1) / 2;
    static_assert(JointLayout<1>::JREC * 2 <= JM::JREC && JointLayout<2>::JREC <= JM::JREC, "pixel records: max_pairs / 4 targets of the S = JMAXS layout hold every case");
    static_assert(JointLayout<1>::NACC * 2 <= JM::NACC && JointLayout<2>::NACC <= JM::NACC, "workgroup records: likewise");
    HIPCHK(h, hipMalloc((void **)&h->jrec, nb * hw * JM::JREC * sizeof(float)));
    HIPCHK(h, hipMalloc((void **)&h->jrec_acc, nb * hw * JM::JREC * sizeof(float)));
    HIPCHK(h, hipMalloc((void **)&h->jdepth_acc, nt * hw * sizeof(float)));
    HIPCHK(h, hipMalloc((void **)&h->jblockrec, nb * 2 * h->nblk_alloc * JM::NACC * sizeof(float)));      // (tile records + the quarter-resolution mode's cell-group records)
    HIPCHK(h, hipMalloc((void **)&h->jstate, nt * sizeof(JointState)));
    HIPCHK(h, hipMalloc((void **)&h->jdelta, nt * 6 * JMAXS * sizeof(double)));
    HIPCHK(h, hipMalloc((void **)&h->jpart, 2 * nt * kJointSplitMax * JM::NACC * sizeof(double)));      // (forward groups, then the free-source mode's inverse groups)
    HIPCHK(h, hipMalloc((void **)&h->jtick, 2 * nt * sizeof(int)));
    HIPCHK(h, hipMemset(h->jtick, 0, 2 * nt * sizeof(int)));
    h->jrec_S = JMAXS;
    return TCSFM_OK;
}
// capacity of that scratch for a call with B targets of NS sources each (explicit: the sizes above are derived, not per call)
template <int NS>
bool joint_scratch_fits(const tcsfm_ctx *h, int B, int recs_per_target) {
    using JL = JointLayout<NS>;
    using JM = JointLayout<JMAXS>;
    const size_t n = h->max_pairs, nb = (n + 3) / 4, nt = (n + 1) / 2;
    return (size_t)B <= nt && (size_t)B * JL::JREC <= nb * JM::JREC &&
           (size_t)B * recs_per_target * JL::NACC <= nb * 2 * (size_t)h->nblk_alloc * JM::NACC;
}

// tile height of k_dense_joint's own grid (32 x 8 tiles of 256 threads by default; 16: the round-4 grid shared with the pair-form kernels)
#ifndef TC_JOINT_TILE_H
#define TC_JOINT_TILE_H 8
#endif

// JOINT dense mode of a window (include/tcsfm.h, tcsfm_refine_dense_window): the S forward pairs of every target share one depth map
// and are solved together (k_dense_joint / k_solve_joint / k_dense_joint_update); the inverse pairs run the pair-form dense kernels on
// offset views of the same scratch.  Inputs already on the device.
template <int NS>
int dense_joint_run(tcsfm_ctx *h, const tcsfm_opts *o, int B, const float *d_tgt, const float *d_src, const float *d_dt, const float *d_ds,
                           const float *d_K, const float *d_pose_in, float *d_pose_out, float *d_depth_out, float *d_stats, const WinOff *wo) {
    using JL = JointLayout<NS>;
    constexpr int S = NS;
    const int SB = S * B, N = 2 * SB;
    const size_t hw = (size_t)h->H * h->W, n = h->max_pairs;
    const bool lm = o->solver == TCSFM_SOLVER_LM;
    const int n_sel = (o->argmin) ? SB : 0;
    int rc;
    constexpr int DTW = 32, DTH = 16, DNT = 512;
    const int tiles_x = (h->W + DTW - 1) / DTW, tiles_y = (h->H + DTH - 1) / DTH, nblk = tiles_x * tiles_y;
    // the joint kernel runs on its own tile grid, as under the reference's loss (third session of round 5): 256-thread workgroups, for S = 2 the
    // LEAN form of the kernel (161 VGPRs: three workgroups per CU instead of one 512-thread workgroup at 212-232)
    constexpr int JTW = 32, JTH = TC_JOINT_TILE_H, JNT = JTW * JTH;
    const int jtiles_x = (h->W + JTW - 1) / JTW, jtiles_y = (h->H + JTH - 1) / JTH, jnblk = jtiles_x * jtiles_y;
    if (n_sel && !h->sel_maps) HIPCHK(h, hipMalloc((void **)&h->sel_maps, (size_t)2 * h->max_pairs * hw * sizeof(float)));
    if (!h->dense_rec) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth0, n * hw * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->delta, n * 8 * sizeof(double)));
    }
    if (lm && !h->dense_rec_acc) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec_acc, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth_acc, n * hw * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->lm_accept, n * sizeof(int)));
    }
    if ((rc = joint_scratch(h))) return rc;
    if (lm && !h->lm_accept) HIPCHK(h, hipMalloc((void **)&h->lm_accept, n * sizeof(int)));
    if ((size_t)nblk > (size_t)h->nblk_alloc || (size_t)jnblk > (size_t)h->nblk_alloc) return fail(h, TCSFM_E_ARG, "internal: dense tile grid exceeds scratch");
    if (!joint_scratch_fits<NS>(h, B, jnblk)) return fail(h, TCSFM_E_ARG, "internal: the joint dense scratch does not hold this many targets");
    tcsfm_opts oo = *o;
    oo.refine = TCSFM_REFINE_POSE;
    oo.window_rule = TCSFM_WINDOW_PAIR;          // (the pose-mode coupling; the joint kernel takes the rule through JointParams)
    InitParams I = init_params(h, &oo, N, d_pose_in, nullptr, d_K, 0);
    I.K_mod = B;
    if ((rc = run_pack(h, &oo, N, d_tgt, d_src, d_dt, d_ds, &I, B, S, h->depth0, wo))) return rc;
    // ---- forward group: joint
    LinParams Pj = lin_params(h, &oo, 6);
    Pj.tiles_x = jtiles_x; Pj.tiles_y = jtiles_y; Pj.ngrp = (jnblk + RG - 1) / RG; Pj.direct = 1;
    float *sel_diff = h->sel_maps, *sel_valid = h->sel_maps ? h->sel_maps + (size_t)h->max_pairs * hw : nullptr;
    if (n_sel) { Pj.ext_diff = sel_diff; Pj.ext_valid = sel_valid; Pj.n_ext = n_sel; Pj.ext_B = B; Pj.ext_S = S; }
    JointParams J;
    memset(&J, 0, sizeof(J));
    J.jrec = h->jrec; J.depth0 = h->depth0; J.jblockrec = h->jblockrec; J.lambda_depth = o->lambda_depth; J.w_prior = o->prior_depth;
    J.B = B; J.S = S; J.argmin = o->argmin ? 1 : 0;
    J.automask = 0;                             // own masks only without argmin, where the reference's forward term has no auto-mask (:71-73)
    JointSolveParams Sj;
    memset(&Sj, 0, sizeof(Sj));
    Sj.jblockrec = h->jblockrec; Sj.js = h->jstate; Sj.st = h->state; Sj.pc = h->pconst; Sj.stats = d_stats; Sj.nblk = jnblk; Sj.B = B;
    Sj.n_iters = o->n_iters; Sj.solver = o->solver; Sj.lambda_up = o->lambda_up; Sj.lambda_down = o->lambda_down; Sj.lambda_min = o->lambda_min;
    Sj.lambda0 = o->lambda0; Sj.delta_out = h->jdelta; Sj.accept_out = lm ? h->lm_accept : nullptr;
    JointUpdateParams Uj;
    memset(&Uj, 0, sizeof(Uj));
    Uj.jrec = h->jrec; Uj.jrec_acc = lm ? h->jrec_acc : nullptr; Uj.depth_acc = lm ? h->jdepth_acc : nullptr; Uj.delta = h->jdelta;
    Uj.accept = lm ? h->lm_accept : nullptr; Uj.depth = h->depth_work; Uj.depth_out = nullptr; Uj.hw = (int)hw; Uj.B = B; Uj.S = S; Uj.mode = 0;
    Uj.rho_lo = 1.f / o->max_depth; Uj.rho_hi = 1.f / o->min_depth;
    // ---- inverse pairs: the pair-form dense kernels on views offset by S B pairs
    LinParams Pi = lin_params(h, &oo, 6);
    Pi.tiles_x = tiles_x; Pi.tiles_y = tiles_y; Pi.ngrp = (nblk + RG - 1) / RG; Pi.direct = 1;
    Pi.tgtpack += (size_t)SB * hw; Pi.srcpack += (size_t)SB * (h->H + 2) * (h->W + 2); Pi.depth_t += (size_t)SB * hw; Pi.pc += SB;
    Pi.blockrec += (size_t)SB * nblk * AccLayout<6>::NACC;
    SolveParams Si = solve_params(h, &oo, 6, 0);
    Si.partials = Pi.blockrec; Si.ngrp = nblk; Si.st = h->state + SB; Si.pc = h->pconst + SB;
    Si.stats = d_stats ? d_stats + (size_t)SB * (o->n_iters + 1) * TCSFM_NSTAT : nullptr;
    Si.delta_out = h->delta + (size_t)SB * 8; Si.accept_out = lm ? h->lm_accept + SB : nullptr;
    DenseParams Dn;
    Dn.dense_rec = h->dense_rec + (size_t)SB * hw * 8; Dn.depth0 = h->depth0 + (size_t)SB * hw; Dn.lambda_depth = o->lambda_depth; Dn.w_prior = o->prior_depth;
    Dn.prev_rec = nullptr; Dn.prev_delta = h->delta; Dn.depth_next = nullptr; Dn.rho_lo = Uj.rho_lo; Dn.rho_hi = Uj.rho_hi;
    DenseUpdateParams Ui;
    memset(&Ui, 0, sizeof(Ui));          // (no coalesce table: c_ncall = 0)
    Ui.dense_rec = Dn.dense_rec; Ui.delta = Si.delta_out; Ui.depth = h->depth_work + (size_t)SB * hw; Ui.depth_out = h->depth_work + (size_t)SB * hw; Ui.hw = (int)hw;
    Ui.rho_lo = Uj.rho_lo; Ui.rho_hi = Uj.rho_hi;
    DenseLmParams Ul;
    Ul.rec_try = Dn.dense_rec; Ul.rec_acc = lm ? h->dense_rec_acc + (size_t)SB * hw * 8 : nullptr; Ul.depth_acc = lm ? h->depth_acc + (size_t)SB * hw : nullptr;
    Ul.depth = h->depth_work + (size_t)SB * hw; Ul.delta = Si.delta_out; Ul.accept = lm ? h->lm_accept + SB : nullptr; Ul.hw = (int)hw; Ul.rho_lo = Uj.rho_lo; Ul.rho_hi = Uj.rho_hi;
    const dim3 px_t((unsigned)((hw + 255) / 256), B), px_i((unsigned)((hw + 255) / 256), SB);
    // The inverse pairs' refinement never meets the forward group's (different pairs, different scratch views): it runs on a second
    // stream beside it -- fork after the pack, join before the results are copied out -- so a call costs max(forward chain, inverse
    // chain) per iteration instead of their sum (80 -> ~50 us per iteration for the KITTI window at 640x192).
    if (!h->aux_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->aux_fork, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->aux_join, hipEventDisableTiming));
    }
    hipStream_t fs = h->stream, is = h->aux_stream;
    const bool tr = h->trace_bits != nullptr;
    if ((rc = trace_check(h, o, N))) return rc;
    HIPCHK(h, hipEventRecord(h->aux_fork, fs));
    HIPCHK(h, hipStreamWaitEvent(is, h->aux_fork, 0));
    // ---- inverse pairs, all iterations, on the second stream
    auto lin_inv = [&](int lin) {
        Pi.trace = tr ? h->trace_bits + ((size_t)lin * N + SB) * hw : nullptr;
        Si.trace_decide = h->trace_decide ? h->trace_decide + (size_t)lin * N + SB : nullptr;
        Pi.stamp = nullptr;
        if (tr) hipLaunchKernelGGL((k_dense_linearize<DTW, DTH, DNT, true>), dim3(nblk, SB), dim3(DNT), 0, is, Pi, Dn);
        else hipLaunchKernelGGL((k_dense_linearize<DTW, DTH, DNT>), dim3(nblk, SB), dim3(DNT), 0, is, Pi, Dn);
    };
    for (int it = 0; it < o->n_iters; it++) {
        lin_inv(it);
        const bool last = !lm && it == o->n_iters - 1;
        Si.it = it; Si.mode = 0; Si.pose_out = last ? d_pose_out + (size_t)SB * 6 : nullptr; Si.log_scale_out = nullptr;
        hipLaunchKernelGGL((k_solve<6>), dim3(SB), dim3(TC_SOLVE_NT), 0, is, Si);
        if (lm) hipLaunchKernelGGL(k_dense_update_lm, px_i, dim3(256), 0, is, Ul);
        else hipLaunchKernelGGL(k_dense_update, px_i, dim3(256), 0, is, Ui);
    }
    if (lm && o->n_iters > 0) {
        lin_inv(o->n_iters);
        Si.it = o->n_iters; Si.mode = 1; Si.pose_out = d_pose_out + (size_t)SB * 6; Si.log_scale_out = nullptr;
        hipLaunchKernelGGL((k_solve<6>), dim3(SB), dim3(TC_SOLVE_NT), 0, is, Si);
        hipLaunchKernelGGL(k_dense_final_lm, px_i, dim3(256), 0, is, (const int *)(h->lm_accept + SB), (const float *)(h->depth_acc + (size_t)SB * hw),
                           h->depth_work + (size_t)SB * hw, (int)hw);
    }
    HIPCHK(h, hipEventRecord(h->aux_join, is));
    // ---- forward group, jointly, on the handle's stream
    auto lin_fwd = [&](int lin) {
        Pj.trace = tr ? h->trace_bits + (size_t)lin * N * hw : nullptr;
        Sj.trace_decide = h->trace_decide ? h->trace_decide + (size_t)lin * N : nullptr;
        if (n_sel) {       // selection masks of the forward pairs at the current poses and the current SHARED depth
            LinParams M = lin_params(h, &oo, 6);
            M.o_diff = sel_diff; M.o_valid = sel_valid;
            launch_lin(h, M, n_sel, 6, false, MODE_MAPS, 2);      // (the selection itself: ext_selected, inside the joint kernel)
        }
        take_stamp(h, Pj, (size_t)jnblk * B);
        ProfScope prof(h, 0);
        if (tr) hipLaunchKernelGGL((k_dense_joint<NS, JTW, JTH, JNT, true>), dim3(jnblk, B), dim3(JNT), 0, fs, Pj, J);
        else hipLaunchKernelGGL((k_dense_joint<NS, JTW, JTH, JNT>), dim3(jnblk, B), dim3(JNT), 0, fs, Pj, J);
    };
    constexpr int SOLVE_NT = JSOLVE_NT;
    for (int it = 0; it < o->n_iters; it++) {
        lin_fwd(it);
        const bool last = !lm && it == o->n_iters - 1;
        Sj.it = it; Sj.mode = 0; Sj.pose_out = last ? d_pose_out : nullptr;
        hipLaunchKernelGGL((k_solve_joint<NS>), dim3(B), dim3(SOLVE_NT), 0, fs, Sj);
        hipLaunchKernelGGL((k_dense_joint_update<NS>), px_t, dim3(256), 0, fs, Uj);
    }
    if (lm && o->n_iters > 0) {
        lin_fwd(o->n_iters);
        Sj.it = o->n_iters; Sj.mode = 1; Sj.pose_out = d_pose_out;
        hipLaunchKernelGGL((k_solve_joint<NS>), dim3(B), dim3(SOLVE_NT), 0, fs, Sj);
        Uj.mode = 1;
        hipLaunchKernelGGL((k_dense_joint_update<NS>), px_t, dim3(256), 0, fs, Uj);
    }
    HIPCHK(h, hipStreamWaitEvent(fs, h->aux_join, 0));
    HIPCHK(h, hipGetLastError());
    if (o->n_iters == 0) {
        FinishParams F;
        F.st = h->state; F.pose_out = d_pose_out; F.log_scale_out = nullptr; F.N = N;
        hipLaunchKernelGGL(k_finish, dim3((N + 63) / 64), dim3(64), 0, h->stream, F);
    }
    HIPCHK(h, hipMemcpyAsync(d_depth_out, h->depth_work, N * hw * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return TCSFM_OK;
}


// the fixed-point scatter sums of the reference-loss dense mode are zero between calls (their consumer clears them); after a fresh
// allocation, an export or a call that failed midway they are cleared here
int dref_clean(tcsfm_ctx *h) {
    if (!h->dref_dirty) return TCSFM_OK;
    const size_t hw = (size_t)h->H * h->W, n = h->max_pairs;
    if (h->dref_ext) HIPCHK(h, hipMemsetAsync(h->dref_ext, 0, ((n + 1) / 2) * hw * 2 * sizeof(long long), h->stream));
    if (h->dref_ext_src) HIPCHK(h, hipMemsetAsync(h->dref_ext_src, 0, (n / 2) * hw * 2 * sizeof(long long), h->stream));
    if (h->jtick) HIPCHK(h, hipMemsetAsync(h->jtick, 0, 2 * ((n + 1) / 2) * sizeof(int), h->stream));      // (a call that failed midway may have left tickets behind)
    h->dref_dirty = false;
    return TCSFM_OK;
}

// tiny export kernel of tcsfm_linearize_dense_window: d loss / d rho = a_f x the joint kernel's per-pixel record
__global__ __launch_bounds__(256) void k_dref_export_grho(const float *jrec, int jrec_stride, const double *exp_out, int exp_stride, float *out, int hw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (idx >= hw) return;
    out[(size_t)b * hw + idx] = (float)(exp_out[(size_t)b * exp_stride + 1] * (double)jrec[((size_t)b * hw + idx) * jrec_stride]);
}

// Dense window mode on the REFERENCE's loss (include/tcsfm.h, dense_ref_kernel.h).  Inputs on the device.  lin_export != nullptr: ONE
// linearisation at the given poses, nothing updated: host outputs of tcsfm_linearize_dense_window.
struct DrefExport { double *scal, *g_pose; float *d_g_rho; const float *d_depth0; float *d_g_rho_src; /* [S B][H W] or null: tcsfm_linearize_dense_window_sources */ };
template <int NS>
int dense_ref_run(tcsfm_ctx *h, const tcsfm_opts *o, int B, const float *d_tgt, const float *d_src, const float *d_dt, const float *d_ds,
                  const float *d_K, const float *d_pose_in, float *d_pose_out, float *d_depth_out, float *d_stats, const WinOff *wo, const DrefExport *ex,
                  const CoalTab *ct = nullptr, float *const *ct_pose = nullptr, float *const *ct_depth = nullptr) {
    using JL = JointLayout<NS>;
    using JM = JointLayout<JMAXS>;
    constexpr int S = NS;
    // B: targets of the launch sequence.  Coalesced calls (ct): ncall queued calls of cB targets each run as ONE sequence over B = ncall cB
    // targets; the loss couples the windows of ONE call (batch normalisers, means over the call's maps): every call is a normaliser group of
    // its own (norm_B = cB targets), the per-map weights take the CALL's batch size Bc, inputs and outputs go through the pointer table.
    const int Bc = ct ? ct->cB : B, ngroups = ct ? ct->ncall : 1, norm_B = ct ? ct->cB : 0;
    const int SB = S * B, N = 2 * SB;
    const size_t hw = (size_t)h->H * h->W, n = h->max_pairs;
    int rc;
    if (ct && (ex || o->free_source_depths != 0 || o->n_iters < 1 || d_stats || h->trace_bits)) return fail(h, TCSFM_E_ARG, "internal: merged reference-loss calls are plain refinements with fixed source maps");
    // The joint kernel's own tile grid (round 5): 32 x 8 tiles of 256 threads by default.  At 256 VGPRs the kernel holds 8 waves per CU either
    // way; as TWO independent 4-wave workgroups their barriers and load phases no longer coincide (one workgroup's gathers run under the
    // other's arithmetic), which a single 8-wave workgroup cannot do -- at the price of a larger halo share (TC_JOINT_TILE_H=16: the round-4 grid).
    constexpr int DTW = 32, DTH = TC_JOINT_TILE_H, DNT = DTW * DTH;
    const int jtiles_x = (h->W + DTW - 1) / DTW, jtiles_y = (h->H + DTH - 1) / DTH;
    const int nblk = jtiles_x * jtiles_y;              // workgroup records per target of the joint kernel
    const int nblk_lin = h->nblk;                      // ... and of k_linearize (the inverse pairs' systems, the FRONT launch)
    if (nblk > h->nblk_alloc) return fail(h, TCSFM_E_ARG, "internal: the joint kernel's tile grid exceeds the scratch");
    const bool sel = o->argmin && S > 1, dc = o->w_dc > 0.f;
    if (!h->sel_maps) HIPCHK(h, hipMalloc((void **)&h->sel_maps, (size_t)2 * h->max_pairs * hw * sizeof(float)));
    if (!h->dense_rec) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth0, n * hw * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->delta, n * 8 * sizeof(double)));
    }
    if ((rc = joint_scratch(h))) return rc;
    if (!h->dref_norms) {
        HIPCHK(h, hipMalloc((void **)&h->dref_norms, 2 * TC_MAX_COAL * sizeof(int)));             // (one (K_f, K_i) pair per normaliser group)
        HIPCHK(h, hipMalloc((void **)&h->dref_ext, ((n + 1) / 2) * hw * 2 * sizeof(long long)));  // (targets <= max_pairs / 2; two sums per pixel)
        HIPCHK(h, hipMalloc((void **)&h->dref_export, ((n + 1) / 2) * (2 + 6 * JMAXS) * sizeof(double)));
        HIPCHK(h, hipMalloc((void **)&h->dref_smooth, ((n + 1) / 2) * 2 * sizeof(float)));
    }
    // the reference's parametrisation (optimizer.py:194-198, 235-239): quarter-resolution unknown, x4 bilinear upsampling (dense_ref_kernel.h)
    const bool qres = !ex && o->depth_param == TCSFM_DEPTH_QUARTER;
    const int nq = (h->H / 4) * (h->W / 4), nqblk = qres ? (nq + QRES_CELLS_PER_WG - 1) / QRES_CELLS_PER_WG : 0;
    if (qres && !h->qres_rho) {
        const size_t nb = (n + 1) / 2;
        HIPCHK(h, hipMalloc((void **)&h->qres_rho, 2 * nb * nq * sizeof(float)));       // (two buffers: k_qres_step_up ping-pongs)
        HIPCHK(h, hipMalloc((void **)&h->qres_rec, nb * nq * JM::JREC * sizeof(float)));
    }
    if (qres && nblk + nqblk > 2 * h->nblk_alloc) return fail(h, TCSFM_E_ARG, "internal: quarter-resolution records exceed the scratch");
    if (!joint_scratch_fits<NS>(h, B, nblk + nqblk)) return fail(h, TCSFM_E_ARG, "internal: the joint dense scratch does not hold this many targets");
    // opts.free_source_depths: the inverse pairs as S B groups of one source (the joint kernel / solve / update on views offset by S B pairs)
    const bool free_src = !ex && o->free_source_depths != 0;
    if (free_src && !h->jrec_src) {
        const size_t ng = n / 2;          // groups <= max_pairs / 2
        HIPCHK(h, hipMalloc((void **)&h->jrec_src, ng * hw * JointLayout<1>::JREC * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->jstate_src, ng * sizeof(JointState)));
        HIPCHK(h, hipMalloc((void **)&h->jdelta_src, ng * 6 * JMAXS * sizeof(double)));
        HIPCHK(h, hipMalloc((void **)&h->dref_ext_src, ng * hw * 2 * sizeof(long long)));
        h->dref_dirty = true;
    }
    if (free_src && qres && !h->qres_rho_src) {
        const size_t ng = n / 2;
        HIPCHK(h, hipMalloc((void **)&h->qres_rho_src, 2 * ng * nq * sizeof(float)));      // (two buffers: k_qres_step_up ping-pongs)
        HIPCHK(h, hipMalloc((void **)&h->qres_rec_src, ng * nq * JointLayout<1>::JREC * sizeof(float)));
    }
    // (the inverse groups' workgroup records follow the forward groups' in jblockrec: both joint launches of a linearisation run before the solve)
    const size_t jrec2_off = (size_t)B * (nblk + nqblk) * JL::NACC;
    if (free_src && jrec2_off + (size_t)SB * (nblk + nqblk) * JointLayout<1>::NACC > ((n + 3) / 4) * 2 * (size_t)h->nblk_alloc * JM::NACC)
        return fail(h, TCSFM_E_ARG, "internal: the inverse groups' records exceed the scratch");
    tcsfm_opts oo = *o;
    oo.refine = TCSFM_REFINE_POSE;
    oo.window_rule = TCSFM_WINDOW_PAIR;          // (the couplings are set explicitly below)
    oo.w_pose_consist = 0.f;                     // (... and so is l_pose_consist: `pc` below)
    const bool pc = o->w_pose_consist > 0.f && !ex;      // optimizer.py:95-96 as a term of this mode (the export of one linearisation leaves it out)
    InitParams I = init_params(h, &oo, N, d_pose_in, nullptr, d_K, 0);
    I.K_mod = B;
    if (pc) I.pose_lin = h->pose_lin;            // buffer 0 of [2][N][12]: every pair's transform at the first linearisation
    // the pack also zeroes the batch counters and -- plain refinement with fixed source maps -- leaves the inputs in the caller's depth
    // output: its inverse slots (the source maps) are final, the forward slots are overwritten by the last back-substitution
    const bool direct_out = !ex && !(o->free_source_depths != 0) && o->n_iters > 0 && (d_depth_out != nullptr || ct != nullptr);
    if ((rc = run_pack(h, &oo, N, d_tgt, d_src, d_dt, d_ds, &I, ct ? 0 : B, S, h->depth0, wo, ct, direct_out && !ct ? d_depth_out : nullptr, h->dref_norms, 2 * TC_MAX_COAL,
                       ct ? ct_depth : nullptr))) return rc;
    if (ex && ex->d_depth0)       // the prior's centre given explicitly (slots of the forward pairs (0, b): index b)
        HIPCHK(h, hipMemcpyAsync(h->depth0, ex->d_depth0, (size_t)B * hw * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if ((rc = trace_check(h, o, N))) return rc;
    const bool tr = h->trace_bits != nullptr;
    float *maps_diff = h->sel_maps, *maps_valid = h->sel_maps + (size_t)h->max_pairs * hw;
    // ---- residual maps of all pairs
    LinParams M = lin_params(h, &oo, 6);
    M.o_diff = maps_diff; M.o_valid = maps_valid;
    // ---- prepass: counts and the scatter of the inverse pairs' depth samples
    LinParams Pp = lin_params(h, &oo, 6);
    Pp.ext_diff = maps_diff; Pp.ext_valid = maps_valid; Pp.n_ext = SB; Pp.ext_B = B; Pp.ext_S = S;
    DrefPrepassParams Dp;
    Dp.diff = maps_diff; Dp.valid = maps_valid; Dp.norms = h->dref_norms; Dp.ext = h->dref_ext; Dp.B = B; Dp.S = S;
    Dp.argmin = o->argmin ? 1 : 0; Dp.automask = o->automask; Dp.eps = o->irls_eps;
    Dp.b_dc = o->w_dc / ((float)(S * Bc) * (float)hw);
    Dp.plain_dif = (!ex && o->free_source_depths != 0) ? 1 : 0;
    // ---- inverse pairs: pose kernels on views offset by S B pairs, window rule REFERENCE (all of them are the rule's inverse group)
    LinParams Pi = lin_params(h, &oo, 6);
    const int nacc6 = AccLayout<6>::NACC;
    Pi.tgtpack += (size_t)SB * hw; Pi.srcpack += (size_t)SB * (h->H + 2) * (h->W + 2); Pi.depth_t += (size_t)SB * hw; Pi.pc += SB;
    Pi.blockrec += (size_t)SB * h->nblk * nacc6; Pi.tickets += (size_t)SB * h->ngrp; Pi.partials += (size_t)SB * h->ngrp * nacc6;
    SolveParams Si = solve_params(h, &oo, 6, 0);
    Si.partials = direct_records(h) ? Pi.blockrec : Pi.partials;
    Si.st = h->state + SB; Si.pc = h->pconst + SB; Si.lin_out = h->lin_out + (size_t)SB * kLinOut;
    Si.rule = 1; Si.grp_fwd = 0; Si.n_pairs = SB; Si.scale_fwd = 1.0; Si.scale_inv = 0.25;
    Si.b_dc = (double)o->w_dc / ((double)(S * Bc) * (double)hw);
    Si.stats = d_stats ? d_stats + (size_t)SB * (o->n_iters + 1) * TCSFM_NSTAT : nullptr;
    // ---- forward group: the joint kernel under the reference's rule
    LinParams Pj = lin_params(h, &oo, 6);
    Pj.tiles_x = jtiles_x; Pj.tiles_y = jtiles_y; Pj.ngrp = (nblk + RG - 1) / RG; Pj.direct = 1;
    if (sel) { Pj.ext_diff = maps_diff; Pj.ext_valid = maps_valid; Pj.n_ext = SB; Pj.ext_B = B; Pj.ext_S = S; }
    JointParams J;
    memset(&J, 0, sizeof(J));
    J.jrec = h->jrec; J.depth0 = h->depth0; J.jblockrec = h->jblockrec; J.w_prior = 0.f;
    J.lambda_depth = ex ? 1e20f : o->lambda_depth;      // (export: depth block frozen, the reduced right-hand side IS the pose gradient)
    J.B = B; J.S = S; J.argmin = o->argmin ? 1 : 0;
    J.automask = o->argmin ? o->automask : 0;     // own masks: with one source the min is the source itself; without argmin no auto-mask (:71-73)
    J.norms = h->dref_norms; J.norm_B = norm_B; J.ext2 = h->dref_ext; J.c_f = o->argmin ? 1.f : 0.25f;
    J.dbg = h->dbg_stamps; Si.dbg = nullptr;       // (diagnostic stamps: the joint kernel's phases in this mode, not the pair solve's)
    J.b_dc = o->w_dc / ((float)(S * Bc) * (float)hw); J.w_init_px = o->prior_init / ((float)Bc * (float)hw);
    J.sig_lo = 1.f / o->max_depth; J.sig_ir = 1.f / (1.f / o->min_depth - 1.f / o->max_depth);
    DrefSmoothParams Ds;
    memset(&Ds, 0, sizeof(Ds));
    const bool smooth = o->w_smooth > 0.f && h->H > 1 && h->W > 1;
    if (smooth) {       // optimizer.py:92-93: l_smooth_weight x [mean over B H (W-1) x-edges + mean over B (H-1) W y-edges]
        J.smooth = h->dref_smooth;
        J.w_smooth_x = o->w_smooth / ((float)Bc * (float)h->H * (float)(h->W - 1)); J.w_smooth_y = o->w_smooth / ((float)Bc * (float)(h->H - 1) * (float)h->W);
        Ds.depth = h->depth_work; Ds.tgtpack = h->tgtpack; Ds.out = h->dref_smooth; Ds.H = h->H; Ds.W = h->W;
        Ds.sig_lo = J.sig_lo; Ds.sig_ir = J.sig_ir; Ds.wx = J.w_smooth_x; Ds.wy = J.w_smooth_y;
    }
    JointSolveParams Sj;
    memset(&Sj, 0, sizeof(Sj));
    {   // TCSFM_DEBUG_STAMPS=2: the phases of target 0's joint SOLVE instead (scripts/diag/solve_front_stamps.py)
        static const int which = [] { const char *e = getenv("TCSFM_DEBUG_STAMPS"); return e ? atoi(e) : 0; }();
        if (which == 2) { Sj.dbg = h->dbg_stamps; J.dbg = nullptr; }
    }
    Sj.jblockrec = h->jblockrec; Sj.js = h->jstate; Sj.st = h->state; Sj.pc = h->pconst; Sj.stats = d_stats; Sj.nblk = nblk; Sj.B = B;
    Sj.n_iters = o->n_iters; Sj.solver = TCSFM_SOLVER_GN; Sj.lambda_up = o->lambda_up; Sj.lambda_down = o->lambda_down; Sj.lambda_min = o->lambda_min;
    Sj.lambda0 = o->lambda0; Sj.delta_out = h->jdelta; Sj.accept_out = nullptr;
    Sj.norms = h->dref_norms; Sj.norm_B = norm_B; Sj.c_f = J.c_f;
    if (ct) {
        Sj.c_ncall = ct->ncall; Sj.c_B = ct->cB; Sj.c_S = ct->cS;
        Si.c_ncall = ct->ncall; Si.c_B = ct->cB; Si.c_S = ct->cS; Si.c_n0 = SB;
        for (int i = 0; i < ct->ncall; i++) { Sj.c_pose_out[i] = ct_pose[i]; Si.c_pose_out[i] = ct_pose[i]; Si.c_ls_out[i] = nullptr; }
    }
    JointUpdateParams Uj;
    memset(&Uj, 0, sizeof(Uj));
    Uj.jrec = h->jrec; Uj.delta = h->jdelta; Uj.depth = h->depth_work; Uj.hw = (int)hw; Uj.B = B; Uj.S = S; Uj.mode = 0;
    Uj.rho_lo = 1.f / o->max_depth; Uj.rho_hi = 1.f / o->min_depth;
    Uj.srcpack_inv = h->srcpack + (size_t)SB * (h->H + 2) * (h->W + 2); Uj.W = h->W; Uj.H = h->H;
    const dim3 px_t((unsigned)((hw + 255) / 256), B), px_all((unsigned)((hw + 255) / 256), N);
    hipStream_t st = h->stream;
    QresParams Q;
    memset(&Q, 0, sizeof(Q));
    if (qres) {
        J.qres = 1; J.rec_stride = nblk + nqblk; Sj.nblk = nblk + nqblk;
        Q.jrec = h->jrec; Q.qrec = h->qres_rec; Q.rho_q = h->qres_rho; Q.jblockrec = h->jblockrec; Q.delta = h->jdelta; Q.depth = h->depth_work;
        Q.srcpack_inv = Uj.srcpack_inv; Q.H = h->H; Q.W = h->W; Q.B = B; Q.S = S; Q.rec_stride = nblk + nqblk; Q.rec_first = nblk;
        Q.rho_lo = Uj.rho_lo; Q.rho_hi = Uj.rho_hi;
        // the start is the quarter-resolution projection of the input map, upsampled again (optimizer.py:194-196, 235); the prior's centre
        // stays the full-resolution input (`self.target_disparity`, :89-90)
        hipLaunchKernelGGL(k_qres_init, dim3((unsigned)((nq + 255) / 256), B), dim3(256), 0, st, Q);
        hipLaunchKernelGGL(k_qres_upsample, px_t, dim3(256), 0, st, Q);
        Q.norms_zero = h->dref_norms; Q.norms_n = 2 * ngroups;
    }
    QresParams Q2 = Q;
    if (qres && free_src) {      // the source maps in the same parametrisation: their quarter-resolution projection is the start (optimizer.py:194-196)
        Q2.jrec = h->jrec_src; Q2.qrec = h->qres_rec_src; Q2.rho_q = h->qres_rho_src; Q2.delta = h->jdelta_src;
        Q2.depth = h->depth_work + (size_t)SB * hw; Q2.srcpack_inv = h->srcpack; Q2.B = SB; Q2.S = 1; Q2.norms_zero = nullptr;
        hipLaunchKernelGGL(k_qres_init, dim3((unsigned)((nq + 255) / 256), SB), dim3(256), 0, st, Q2);
        hipLaunchKernelGGL(k_qres_upsample, dim3((unsigned)((hw + 255) / 256), SB), dim3(256), 0, st, Q2);
    }
    // (one stream: running the inverse pairs' linearise + solve on a second stream beside the forward group's, forked behind the scatter and
    // joined after the depth update, was measured SLOWER -- 261 vs 232 us per 240x320 window, 383 vs 361 at 192x640 S=2: the event hops cost
    // more than the ~18 us of overlap they buy)
    // the counters and the scatter sums are zero when a linearisation starts: zeroed here once per call, and by their consumers afterwards
    // (k_dense_joint clears every sum it reads, k_dense_joint_update the counters) -- two memset launches per iteration cost 10 us
    if ((rc = dref_clean(h))) return rc;
    h->dref_dirty = true;          // (until this call has issued its last consumer)
    Uj.norms_zero = h->dref_norms; Uj.norms_n = 2 * ngroups;
    LinParams Pj2 = lin_params(h, &oo, 6);
    JointParams J2 = J;
    JointSolveParams Sj2 = Sj;
    JointUpdateParams Uj2 = Uj;
    if (free_src) {
        Pj2.tgtpack += (size_t)SB * hw; Pj2.srcpack += (size_t)SB * (h->H + 2) * (h->W + 2); Pj2.depth_t += (size_t)SB * hw; Pj2.pc += SB;
        Pj2.tiles_x = jtiles_x; Pj2.tiles_y = jtiles_y; Pj2.ngrp = (nblk + RG - 1) / RG; Pj2.direct = 1;
        J2.jrec = h->jrec_src; J2.depth0 = nullptr; J2.B = SB; J2.S = 1; J2.argmin = 0; J2.automask = o->automask;
        J2.norms = h->dref_norms + 1;                 // the group's normaliser is K_i, its factor 0.25 (optimizer.py:79)
        J2.c_f = 0.25f; J2.w_init_px = 0.f; J2.smooth = nullptr; J2.w_smooth_x = J2.w_smooth_y = 0.f; J2.qres = 0; J2.rec_stride = 0;
        J2.ext2 = h->dref_ext_src; J2.ext_norm = h->dref_norms; J2.ext_c = J.c_f;
        J2.jblockrec = h->jblockrec + jrec2_off; Sj2.jblockrec = h->jblockrec + jrec2_off; Q2.jblockrec = h->jblockrec + jrec2_off;
        if (qres) { J2.qres = 1; J2.rec_stride = nblk + nqblk; }
        Sj2.js = h->jstate_src; Sj2.st = h->state + SB; Sj2.pc = h->pconst + SB; Sj2.B = SB; Sj2.nblk = qres ? nblk + nqblk : nblk;
        Sj2.stats = d_stats ? d_stats + (size_t)SB * (o->n_iters + 1) * TCSFM_NSTAT : nullptr;
        Sj2.delta_out = h->jdelta_src; Sj2.norms = h->dref_norms + 1; Sj2.c_f = 0.25;
        Uj2.jrec = h->jrec_src; Uj2.delta = h->jdelta_src; Uj2.depth = h->depth_work + (size_t)SB * hw; Uj2.B = SB; Uj2.S = 1;
        Uj2.norms_zero = nullptr;                     // (the forward update zeroes the counters: it runs after both solves)
        Uj2.srcpack_inv = h->srcpack;                 // forward pair m samples source map m: the depth channel of ITS pack
    }
    // Round 5: with the source maps fixed a linearisation OPENS with one launch over all 2 S B pairs (k_linearize<FRONT>: the forward pairs'
    // selection and K_f, the inverse pairs' systems, K_i and their adjoint scatter) in place of the residual-map, count, scatter and
    // inverse-linearisation launches; the free-source-map mode keeps those (its inverse pairs are linearised by the joint kernel).
    // (second session: the free-source-map mode opens with the same launch -- every row light there, the forward rows scattering the adjoint
    // of their samples of the source maps -- and its two joint groups share ONE solve launch and ONE update launch)
    const bool front = true;
    LinParams F = lin_params(h, &oo, 6);
    if (front) {
        F.front_fwd = SB; F.front_Bt = B; F.norm_B = norm_B; F.norms = h->dref_norms; F.ext2 = h->dref_ext;
        F.front_light = free_src ? 1 : 0; F.ext2_src = h->dref_ext_src;
        F.fwd_noauto = o->argmin ? 0 : SB;                       // optimizer.py:71-73: without argmin the forward term has no auto-mask
        F.one_generation = (size_t)nblk_lin * (SB + (sel ? B : SB)) <= 512;
        if (sel) { F.sel_B = B; F.sel_S = S; F.sel_out = maps_valid; Pj.ext_diff = nullptr; Pj.ext_valid = nullptr; Pj.n_ext = 0; Pj.sel_in = maps_valid; }
        Si.norms = h->dref_norms; Si.norm_Bt = B; Si.norm_B = norm_B;
    }
    auto launch_front = [&](int lin) {
        F.trace = tr ? h->trace_bits + (size_t)lin * N * hw : nullptr;
        F.stamp = nullptr;
        ProfScope prof(h, 2);
        const dim3 grid(nblk_lin, SB + (sel ? B : SB)), block(TILE_NT);     // inverse pairs, then the forward rows (under the selection: one per target)
        if (tr) {
            if (sel) hipLaunchKernelGGL((k_linearize<6, true, MODE_LIN, TILE_W, TILE_H, TILE_NT, true, true, false, true>), grid, block, 0, st, F);
            else hipLaunchKernelGGL((k_linearize<6, true, MODE_LIN, TILE_W, TILE_H, TILE_NT, false, true, false, true>), grid, block, 0, st, F);
        } else if (sel) hipLaunchKernelGGL((k_linearize<6, true, MODE_LIN, TILE_W, TILE_H, TILE_NT, true, false, false, true>), grid, block, 0, st, F);
        else hipLaunchKernelGGL((k_linearize<6, true, MODE_LIN, TILE_W, TILE_H, TILE_NT, false, false, false, true>), grid, block, 0, st, F);
    };
    auto linearise = [&](int lin) -> int {
        Pi.trace = tr ? h->trace_bits + ((size_t)lin * N + SB) * hw : nullptr;
        Si.trace_decide = h->trace_decide ? h->trace_decide + (size_t)lin * N + SB : nullptr;
        launch_front(lin);
        Pj.trace = tr ? h->trace_bits + (size_t)lin * N * hw : nullptr;
        Sj.trace_decide = h->trace_decide ? h->trace_decide + (size_t)lin * N : nullptr;
        if (smooth) hipLaunchKernelGGL(k_dref_smooth, dim3(B), dim3(1024), 0, st, Ds);
        const bool both = free_src && !smooth;       // free source maps: the forward groups and the inverse pairs' groups of one source in ONE launch
        take_stamp(h, Pj, (size_t)nblk * (both ? B + SB : B));
        ProfScope prof(h, 0);
        if (both) {
            Pj2.trace = tr ? h->trace_bits + ((size_t)lin * N + SB) * hw : nullptr;
            Pj2.stamp = Pj.stamp;                    // (stamps are indexed by the launch's grid)
            if (tr) hipLaunchKernelGGL((k_dense_joint2<NS, DTW, DTH, DNT, true>), dim3(nblk, B + SB), dim3(DNT), 0, st, Pj, J, Pj2, J2);
            else hipLaunchKernelGGL((k_dense_joint2<NS, DTW, DTH, DNT, false>), dim3(nblk, B + SB), dim3(DNT), 0, st, Pj, J, Pj2, J2);
            if (qres) hipLaunchKernelGGL((k_qres_schur2<NS>), dim3(nqblk, B + SB), dim3(256), 0, st, Q, Q2);      // both groups' cells
            return TCSFM_OK;
        }
        // (l_smooth is compiled into its own instantiations: without it -- the reference's drivers -- the S = 2 kernel is the LEAN form, three workgroups per CU)
        if (smooth) {
            if (tr) hipLaunchKernelGGL((k_dense_joint<NS, DTW, DTH, DNT, true, true, true>), dim3(nblk, B), dim3(DNT), 0, st, Pj, J);
            else hipLaunchKernelGGL((k_dense_joint<NS, DTW, DTH, DNT, false, true, true>), dim3(nblk, B), dim3(DNT), 0, st, Pj, J);
        } else if (tr) hipLaunchKernelGGL((k_dense_joint<NS, DTW, DTH, DNT, true, true, false>), dim3(nblk, B), dim3(DNT), 0, st, Pj, J);
        else hipLaunchKernelGGL((k_dense_joint<NS, DTW, DTH, DNT, false, true, false>), dim3(nblk, B), dim3(DNT), 0, st, Pj, J);
        if (qres && !free_src) hipLaunchKernelGGL((k_qres_schur<NS>), dim3(nqblk, B), dim3(256), 0, st, Q);
        if (free_src) {
            // every inverse pair as a group of one source WITHOUT argmin: that is the reference's inverse term (0.25 / K_i, own weights,
            // valid x auto-mask, its depth-consistency term), its local unknowns the pair's pose and the source map it back-projects.
            // Independent of the forward groups' launch above (own records behind theirs in jblockrec): one solve launch serves both.
            Pj2.trace = tr ? h->trace_bits + ((size_t)lin * N + SB) * hw : nullptr;
            if (tr) hipLaunchKernelGGL((k_dense_joint<1, DTW, DTH, DNT, true, true, false>), dim3(nblk, SB), dim3(DNT), 0, st, Pj2, J2);
            else hipLaunchKernelGGL((k_dense_joint<1, DTW, DTH, DNT, false, true, false>), dim3(nblk, SB), dim3(DNT), 0, st, Pj2, J2);
            if (qres) hipLaunchKernelGGL((k_qres_schur2<NS>), dim3(nqblk, B + SB), dim3(256), 0, st, Q, Q2);      // both groups' cells
        }
        return TCSFM_OK;
    };
    if (ex) {        // one linearisation, exported
        if ((rc = linearise(0))) return rc;
        Si.mode = 2; Si.it = 0;
        launch_solve(h, Si, SB, 6);
        Sj.it = 0; Sj.mode = 0; Sj.export_out = h->dref_export;
        hipLaunchKernelGGL((k_solve_joint<NS>), dim3(B), dim3(JSOLVE_NT), 0, st, Sj);
        hipLaunchKernelGGL(k_dref_export_grho, px_t, dim3(256), 0, st, (const float *)h->jrec, (int)JL::JREC, (const double *)h->dref_export, 2 + 6 * JMAXS, ex->d_g_rho, (int)hw);
        if (ex->d_g_rho_src) {
            // The gradient w.r.t. the SOURCE inverse-depth maps (held fixed by the refinement; leaves of the reference's optimize_depth_pred).
            // Their role mirrors the target map's: local in the inverse pair -- the joint kernel run on the inverse pairs as S = 1 groups
            // without argmin IS the reference's inverse term (0.25 / K_i, own weights, valid x auto-mask, its depth-consistency term) --
            // and sampled by the forward pair: k_dref_scatter_src, in units of the forward factor a_f.  Oracle: dref_source_depth_gradient.
            HIPCHK(h, hipMemsetAsync(h->dref_ext, 0, (size_t)SB * hw * 2 * sizeof(long long), st));
            const dim3 px_f((unsigned)((hw + 255) / 256), SB);
            launch_lin(h, M, N, 6, false, MODE_MAPS, 2);          // (the forward pairs' residual maps: what k_dref_scatter_src weighs the samples with)
            hipLaunchKernelGGL(k_dref_scatter_src, px_f, dim3(256), 0, st, Pp, Dp, h->dref_ext, J.c_f);
            LinParams Pj2 = lin_params(h, &oo, 6);
            Pj2.tgtpack += (size_t)SB * hw; Pj2.srcpack += (size_t)SB * (h->H + 2) * (h->W + 2); Pj2.depth_t += (size_t)SB * hw; Pj2.pc += SB;
            Pj2.tiles_x = jtiles_x; Pj2.tiles_y = jtiles_y; Pj2.ngrp = (nblk + RG - 1) / RG; Pj2.direct = 1;
            JointParams J2 = J;
            J2.jrec = h->jrec_acc; J2.depth0 = nullptr; J2.B = SB; J2.S = 1; J2.argmin = 0; J2.automask = o->automask;
            J2.norms = h->dref_norms + 1;                 // the group's normaliser is K_i, its factor 0.25 (optimizer.py:79)
            J2.c_f = 0.25f; J2.w_init_px = 0.f; J2.smooth = nullptr; J2.w_smooth_x = J2.w_smooth_y = 0.f; J2.qres = 0; J2.rec_stride = 0;
            J2.ext2 = h->dref_ext; J2.ext_norm = h->dref_norms; J2.ext_c = J.c_f;
            hipLaunchKernelGGL((k_dense_joint<1, DTW, DTH, DNT, false, true, false>), dim3(nblk, SB), dim3(DNT), 0, st, Pj2, J2);
            hipLaunchKernelGGL(k_dref_export_grho_src, px_f, dim3(256), 0, st, (const float *)h->jrec_acc, (int)JointLayout<1>::JREC, (const int *)h->dref_norms, ex->d_g_rho_src, (int)hw);
        }
        HIPCHK(h, hipGetLastError());
        constexpr int kLin6 = 6 * 6 + 6 + 4;          // k_solve<6> mode 2: H [36], g [6], cost, cost_photo, cost_dc, n_mask
        std::vector<double> lin((size_t)SB * kLin6), jx((size_t)B * (2 + 6 * JMAXS));
        int norms[4];
        HIPCHK(h, hipMemcpyAsync(lin.data(), Si.lin_out, lin.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(jx.data(), h->dref_export, jx.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(norms, h->dref_norms, sizeof(norms), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        double fwd = 0, inv_p = 0, inv_d = 0;
        for (int b = 0; b < B; b++) {
            fwd += jx[(size_t)b * (2 + 6 * JMAXS)];
            for (int s_ = 0; s_ < S; s_++)
                for (int j = 0; j < 6; j++) ex->g_pose[(size_t)(s_ * B + b) * 6 + j] = jx[(size_t)b * (2 + 6 * JMAXS) + 2 + 6 * s_ + j];
        }
        for (int m = 0; m < SB; m++) {
            const double *q = &lin[(size_t)m * kLin6];
            for (int j = 0; j < 6; j++) ex->g_pose[(size_t)(SB + m) * 6 + j] = q[36 + j];
            inv_p += q[36 + 6 + 1]; inv_d += q[36 + 6 + 2];
        }
        double pc_total = 0.0;
        if (o->w_pose_consist > 0.f) {
            // l_pose_consist (optimizer.py:95-96) of the exported linearisation: a closed form of the 2 S B input poses, added on the host in double --
            // value c sum |p_fwd + p_inv|, c = w / (6 S B); gradient c r / max(|r|, eps) in pose coordinates carried to the left perturbation
            // by A^-T, A^-1 = [[-I, Tx], [0, -Je^-1]] (the oracle's pose_consist_term; the refinement's kernels hold the same term)
            std::vector<float> ph((size_t)N * 6);
            HIPCHK(h, hipMemcpy(ph.data(), d_pose_in, ph.size() * sizeof(float), hipMemcpyDeviceToHost));
            const double c = (double)o->w_pose_consist / (6.0 * SB), eps = (double)o->irls_eps;
            for (int m = 0; m < N; m++) {
                const float *pm = &ph[(size_t)m * 6], *pp = &ph[(size_t)(m < SB ? m + SB : m - SB) * 6];
                const double cx = cos(-(double)pm[3]), sx = sin(-(double)pm[3]), cy = cos(-(double)pm[4]), sy = sin(-(double)pm[4]);
                const double Je[9] = {1, 0, sy, 0, cx, -sx * cy, 0, sx, cx * cy}, id = 1.0 / cy;
                const double Ji[9] = {(Je[4] * Je[8] - Je[5] * Je[7]) * id, -(Je[1] * Je[8] - Je[2] * Je[7]) * id, (Je[1] * Je[5] - Je[2] * Je[4]) * id,
                                      -(Je[3] * Je[8] - Je[5] * Je[6]) * id, (Je[0] * Je[8] - Je[2] * Je[6]) * id, -(Je[0] * Je[5] - Je[2] * Je[3]) * id,
                                      (Je[3] * Je[7] - Je[4] * Je[6]) * id, -(Je[0] * Je[7] - Je[1] * Je[6]) * id, (Je[0] * Je[4] - Je[1] * Je[3]) * id};
                const double tq[3] = {-(double)pm[0], -(double)pm[1], -(double)pm[2]};
                const double Tx[9] = {0, -tq[2], tq[1], tq[2], 0, -tq[0], -tq[1], tq[0], 0};
                double Ai[36] = {0}, gp[6];
                for (int i = 0; i < 3; i++) {
                    Ai[6 * i + i] = -1;
                    for (int j = 0; j < 3; j++) { Ai[6 * i + 3 + j] = Tx[3 * i + j]; Ai[6 * (3 + i) + 3 + j] = -Ji[3 * i + j]; }
                }
                for (int j = 0; j < 6; j++) {
                    const double r = (double)pm[j] + (double)pp[j], a = fabs(r), den = a > eps ? a : eps;
                    pc_total += 0.5 * c * a;
                    gp[j] = c * r / den;
                }
                for (int i = 0; i < 6; i++) {
                    double v = 0.0;
                    for (int k = 0; k < 6; k++) v += Ai[6 * k + i] * gp[k];
                    ex->g_pose[(size_t)m * 6 + i] += v;
                }
            }
        }
        ex->scal[0] = fwd + inv_p + inv_d + pc_total; ex->scal[1] = fwd; ex->scal[2] = inv_p; ex->scal[3] = inv_d;
        ex->scal[4] = norms[0]; ex->scal[5] = norms[1]; ex->scal[6] = jx[1]; ex->scal[7] = pc_total;
        return TCSFM_OK;
    }
    // the targets' record sums are split over several workgroups (JointSolveParams::nsplit: the last arriver solves) when a target has MANY records
    // -- the quarter-resolution unknown's tile + cell-group records.  Measured (scripts/dense_ref_timing.py, TCSFM_JOINT_SPLIT=1 switches it off):
    // 600-960 records: -2 ... -8 % per call; 150-480 records: the ticket costs what the faster fetch buys (+0 ... +2 %): not split -- except the S >= 2
    // targets' 480 records of 97+ floats, which two workgroups sum in one batch of loads each (third session: -0.8 % per B=1 KITTI window, -2.8 % at
    // minibatch 6; with one source the 32-float records are one batch already and splitting costs 2-4 %: profiles/r05_joint_split_sweep.txt)
    // (after the S = 1 solve took 32 record subsets its 600-900 quarter-resolution records are best summed by ONE workgroup -- 185 -> 175 us per 240x320
    // call -- and the S >= 2 targets' by four: profiles/r05_joint_split_sweep.txt)
    auto split_of = [](int recs, int ns) { return ns == 1 ? 1 : (recs >= 512 ? 4 : (recs >= 400 ? 2 : 1)); };
    static const int split_env = getenv("TCSFM_JOINT_SPLIT") ? atoi(getenv("TCSFM_JOINT_SPLIT")) : -1;       // (A/B hook: 1 = off)
    {
        const size_t nt = (n + 1) / 2;
        Sj.nsplit = split_env > 0 ? std::min(split_env, kJointSplitMax) : split_of(Sj.nblk, NS); Sj.jpart = h->jpart; Sj.jtick = h->jtick;
        Sj2.nsplit = split_env > 0 ? std::min(split_env, kJointSplitMax) : split_of(Sj2.nblk, 1); Sj2.jpart = h->jpart + nt * kJointSplitMax * JM::NACC; Sj2.jtick = h->jtick + nt;
    }
    if (pc) {     // c = weight / (6 S B) of the CALL (merged calls never carry the term); forward pairs at [0, SB), inverse pairs at [SB, 2 SB) of a buffer
        const double c = (double)o->w_pose_consist / (6.0 * SB);
        Sj.w_pc = c; Sj.pc_eps = (double)o->irls_eps; Sj.pose_lin = h->pose_lin; Sj.pc_np = N; Sj.pc_self0 = 0; Sj.pc_part0 = SB;
        Sj2.w_pc = c; Sj2.pc_eps = (double)o->irls_eps; Sj2.pose_lin = h->pose_lin; Sj2.pc_np = N; Sj2.pc_self0 = SB; Sj2.pc_part0 = 0;
        Si.w_pc = c; Si.pc_eps = (double)o->irls_eps; Si.pose_lin = h->pose_lin; Si.pc_np = N; Si.pc_self0 = SB; Si.pc_part0 = 0;
    }
    for (int it = 0; it < o->n_iters; it++) {
        if ((rc = linearise(it))) return rc;
        const bool last = it == o->n_iters - 1;
        Si.it = it; Si.mode = 0; Si.pose_out = last ? d_pose_out + (size_t)SB * 6 : nullptr; Si.log_scale_out = nullptr;
        Sj.it = it; Sj.mode = 0; Sj.pose_out = last ? d_pose_out : nullptr;
        if (free_src) {   // the forward groups' 6S x 6S systems and the inverse groups' (pose + source map) 6 x 6 systems: one launch
            ProfScope prof(h, 1);
            Sj2.it = it; Sj2.mode = 0; Sj2.pose_out = last ? d_pose_out + (size_t)SB * 6 : nullptr;
            Sj2.trace_decide = h->trace_decide ? h->trace_decide + (size_t)it * N + SB : nullptr;
            if (pc) hipLaunchKernelGGL((k_solve_joint2<NS, true>), dim3(B * Sj.nsplit + SB * Sj2.nsplit), dim3(JSOLVE_NT), 0, st, Sj, Sj2);
            else hipLaunchKernelGGL((k_solve_joint2<NS>), dim3(B * Sj.nsplit + SB * Sj2.nsplit), dim3(JSOLVE_NT), 0, st, Sj, Sj2);
        } else {          // the target groups' and the inverse pairs' systems: independent, one launch
            ProfScope prof(h, 1);
            if (pc) hipLaunchKernelGGL((k_solve_front<NS, true>), dim3(B * Sj.nsplit + SB), dim3(JSOLVE_NT), 0, st, Sj, Si);
            else hipLaunchKernelGGL((k_solve_front<NS>), dim3(B * Sj.nsplit + SB), dim3(JSOLVE_NT), 0, st, Sj, Si);
        }
        if (direct_out && last) {      // the last back-substitution also writes the caller's map (coalesced calls: every call's own)
            Uj.depth_out = d_depth_out; Q.depth_out = d_depth_out;
            if (ct) {
                Uj.c_ncall = Q.c_ncall = ct->ncall; Uj.c_B = Q.c_B = ct->cB;
                for (int i = 0; i < ct->ncall; i++) { Uj.c_depth_out[i] = ct_depth[i]; Q.c_depth_out[i] = ct_depth[i]; }
            }
        }
        const unsigned qsu_tiles = (unsigned)(((h->W + QSU_TW - 1) / QSU_TW) * ((h->H + QSU_TH - 1) / QSU_TH));
        if (qres) {       // cell step + x4 upsampling in one launch; the cell values ping-pong between the two halves of qres_rho
            const size_t half = ((n + 1) / 2) * (size_t)nq;
            Q.rho_q = h->qres_rho + (it & 1) * half; Q.rho_q_next = h->qres_rho + ((it + 1) & 1) * half;
            if (free_src) {       // ... and the source maps' cells in the same launch (their upsampling refreshes the packs the forward pairs sample)
                const size_t half2 = (n / 2) * (size_t)nq;
                Q2.rho_q = h->qres_rho_src + (it & 1) * half2; Q2.rho_q_next = h->qres_rho_src + ((it + 1) & 1) * half2;
                hipLaunchKernelGGL((k_qres_step_up2<NS>), dim3(qsu_tiles, B + SB), dim3(QSU_TW * QSU_TH), 0, st, Q, Q2);
            } else hipLaunchKernelGGL((k_qres_step_up<NS>), dim3(qsu_tiles, B), dim3(QSU_TW * QSU_TH), 0, st, Q);
        } else if (free_src)      // the targets' maps and the source maps (the inverse pairs' own depth slots AND the depth channel of the packs the
                                  // forward pairs sample) in one launch
            hipLaunchKernelGGL((k_dense_joint_update2<NS>), dim3((unsigned)((hw + 255) / 256), B + SB), dim3(256), 0, st, Uj, Uj2);
        else hipLaunchKernelGGL((k_dense_joint_update<NS>), px_t, dim3(256), 0, st, Uj);
    }
    HIPCHK(h, hipGetLastError());
    if (o->n_iters == 0) {
        FinishParams F;
        F.st = h->state; F.pose_out = d_pose_out; F.log_scale_out = nullptr; F.N = N;
        hipLaunchKernelGGL(k_finish, dim3((N + 63) / 64), dim3(64), 0, st, F);
    }
    if (!direct_out) HIPCHK(h, hipMemcpyAsync(d_depth_out, h->depth_work, N * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
    h->dref_dirty = o->n_iters == 0;      // (every linearisation's sums were consumed and cleared; with no iteration the pack's zeroed counters are all there is)
    return TCSFM_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

static int flush_pending(tcsfm_ctx *h);
static int join_coalesce_lanes(tcsfm_ctx *h);
// Calls noted by the *_queued entry points are launched -- and the handle's stream ordered behind the merged sequences that ran on lanes --
// before ANY other entry point puts work on the handle's stream, switches that stream or destroys the handle: a plain call never overtakes
// a call queued before it (producers and consumers of the caller's pose / depth buffers keep their program order; include/tcsfm.h).
static int drain_queued(tcsfm_ctx *h) {
    if (!h || (h->pending.empty() && !h->coal_dirty)) return TCSFM_OK;
    DeviceGuard dev_guard(h->device);
    if (int rc = flush_pending(h)) return rc;
    return join_coalesce_lanes(h);
}

void tcsfm_default_opts(tcsfm_opts *o) {
    memset(o, 0, sizeof(*o));
    o->n_iters = 4; o->solver = TCSFM_SOLVER_GN; o->param = TCSFM_PARAM_SE3; o->refine = TCSFM_REFINE_POSE;
    o->automask = 1; o->depth_is_disp = 0; o->host_ptrs = 0;
    o->w_l1 = 0.15f; o->w_ssim = 0.85f; o->w_dc = 0.f; o->irls_eps = 1e-3f;
    o->lambda0 = 1e-4f; o->lambda_up = 10.f; o->lambda_down = 0.1f; o->lambda_min = 1e-5f;
    o->min_depth = 0.06f; o->max_depth = 2.67f;
    o->prior_scale = 1.0f;
    o->lambda_depth = 1.0f; o->prior_depth = 10.0f;
    o->window_rule = TCSFM_WINDOW_PAIR; o->dense_joint = 1;
    o->prior_init = 0.1f;
    o->depth_param = TCSFM_DEPTH_FULL;
    o->w_pose_consist = 0.f;
    o->w_smooth = 0.f;
    o->free_source_depths = 0;
}

int tcsfm_algorithmic_bytes_per_pixel(const tcsfm_opts *) { return 32; }

const char *tcsfm_last_error(tcsfm_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tcsfm_create(tcsfm_handle *out, int device, int H, int W, int max_pairs) {
    if (!out) return TCSFM_E_ARG;
    *out = nullptr;
    // (the warp gathers address a bordered image with 32-bit byte offsets -- tap4_fetch: (H + 2)(W + 2) 16 B must stay below 4 GiB)
    if (H < 4 || W < 4 || H > 16384 || W > 16384 || max_pairs < 1 || (size_t)H * W * max_pairs > ((size_t)1 << 33) ||
        (size_t)(H + 2) * (W + 2) >= ((size_t)1 << 28)) {
        g_create_error = "tcsfm_create: bad sizes";
        return TCSFM_E_ARG;
    }
    tcsfm_ctx *h = new tcsfm_ctx();
    h->device = device; h->H = H; h->W = W; h->max_pairs = max_pairs;
    h->tiles_x = (W + TILE_W - 1) / TILE_W; h->tiles_y = (H + TILE_H - 1) / TILE_H; h->nblk = h->tiles_x * h->tiles_y;
    h->ngrp = (h->nblk + RG - 1) / RG;
    h->nblk_alloc = ((W + 15) / 16) * ((H + 15) / 16);   // the finest tiling any kernel uses (dense mode, 16x16)
    if (h->nblk_alloc < h->nblk) h->nblk_alloc = h->nblk;
    h->ngrp_alloc = (h->nblk_alloc + RG - 1) / RG;
    h->ngrp_pad = (h->ngrp_alloc + 63) / 64 * 64;
    size_t hw = (size_t)H * W, n = max_pairs;
    DeviceGuard dev_guard(device);
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != device) e = hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->err_host, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { *h->err_host = 0; e = hipHostGetDevicePointer((void **)&h->err_dev, h->err_host, 0); }
    h->stream = h->own_stream;
    if (e == hipSuccess) e = hipMalloc((void **)&h->tgtpack, n * hw * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void **)&h->srcpack, n * (size_t)(H + 2) * (W + 2) * sizeof(float4));   // zero-bordered
    if (e == hipSuccess) e = hipMalloc((void **)&h->depth_work, n * hw * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->partials, n * h->ngrp_pad * kMaxAcc * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->partials, 0, n * h->ngrp_pad * kMaxAcc * sizeof(float));  // pad records stay 0
    if (e == hipSuccess) e = hipMalloc((void **)&h->blockrec, n * h->nblk_alloc * kMaxAcc * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->tickets, n * h->ngrp_alloc * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->tickets, 0, n * h->ngrp_alloc * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->state, n * sizeof(PairState));
    if (e == hipSuccess) e = hipMalloc((void **)&h->pconst, n * sizeof(PairConst));
    if (e == hipSuccess) e = hipMalloc((void **)&h->lin_out, n * kLinOut * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->pose_dev, n * 6 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ls_dev, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->K_dev, n * 9 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->pair_idx, 2 * n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->pose_lin, 2 * n * 12 * sizeof(double));
    if (e != hipSuccess) {
        g_create_error = std::string("tcsfm_create: ") + hipGetErrorString(e);
        tcsfm_destroy(h);
        return e == hipErrorOutOfMemory ? TCSFM_E_NOMEM : TCSFM_E_HIP;
    }
    if (getenv("TCSFM_DEBUG_STAMPS")) {
        (void)hipMalloc((void **)&h->dbg_stamps, 8 * sizeof(long long));
        (void)hipMemset(h->dbg_stamps, 0, 8 * sizeof(long long));
    }
    *out = h;
    return TCSFM_OK;
}

void tcsfm_destroy(tcsfm_handle h) {
    if (!h) return;
    (void)drain_queued(h);                // queued calls are not dropped: they run, and the synchronisations below wait for them
    for (tcsfm_ctx *c : h->lanes) tcsfm_destroy(c);
    h->lanes.clear();
    DeviceGuard dev_guard(h->device);
    if (h->in_ev) (void)hipEventDestroy(h->in_ev);
    if (h->done_ev) (void)hipEventDestroy(h->done_ev);
    for (auto &e : h->marks) (void)hipEventDestroy(e);
    if (h->seq_copy) { (void)hipStreamSynchronize(h->seq_copy); (void)hipStreamDestroy(h->seq_copy); }
    if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
    if (h->aux_fork) (void)hipEventDestroy(h->aux_fork);
    if (h->aux_join) (void)hipEventDestroy(h->aux_join);
    if (h->seq_pack) { (void)hipStreamSynchronize(h->seq_pack); (void)hipStreamDestroy(h->seq_pack); }
    if (!h->graphs.empty()) {             // a replay may still be running: drain the streams it can be on before the executables go
        if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
        if (h->stream && h->stream != h->own_stream) (void)hipStreamSynchronize(h->stream);
        for (auto &g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        h->graphs.clear();
    }
    for (auto &e : h->seq_copied) (void)hipEventDestroy(e);
    for (auto &e : h->seq_raw) (void)hipEventDestroy(e);
    for (auto &e : h->seq_done) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    void *ptrs[] = {h->stamp_buf, h->tgtpack, h->srcpack, h->depth_work, h->partials, h->blockrec, h->tickets, h->state, h->pconst, h->lin_out,
                    h->jrec, h->jrec_acc, h->jblockrec, h->jdepth_acc, h->jstate, h->jdelta, h->jpart, h->jtick, h->dref_norms, h->dref_ext, h->dref_export, h->qres_rho, h->qres_rec, h->pose_lin, h->dref_smooth, h->jrec_src, h->jstate_src, h->jdelta_src, h->dref_ext_src, h->qres_rho_src, h->qres_rec_src,
                    h->pose_dev, h->ls_dev, h->K_dev, h->stats_dev, h->dense_rec, h->depth0, h->dense_rec2, h->depth_alt, h->delta, h->scale_keys, h->scale_hist, h->sel_maps, h->dense_rec_acc, h->depth_acc, h->lm_accept,
                    h->seq_fpack, h->seq_fdepth, h->pair_idx, h->seq_img, h->seq_depth, h->seq_pose_in, h->seq_pose_out, h->seq_ls_out, h->seq_K, h->seq_dense, h->seq_dense_tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &s : h->stage)
        if (s.p) (void)hipFree(s.p);
    for (auto &e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->err_host) (void)hipHostFree(h->err_host);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int tcsfm_set_stream(tcsfm_handle h, void *hip_stream) {
    if (!h) return TCSFM_E_ARG;
    if ((hipStream_t)hip_stream != h->stream)           // queued calls were noted while bound to the old stream: they run there, behind its producers
        if (int rc_q = drain_queued(h)) return rc_q;
    if ((hipStream_t)hip_stream != h->stream && !h->graphs.empty()) {   // captured calls may still be running on the old stream, and a graph
        DeviceGuard dev_guard(h->device);                              // captured there is not replayed on another producer's stream
        drop_graphs(h);
    }
    h->stream = (hipStream_t)hip_stream;  // NULL = legacy default stream
    k_forget(h);                          // a new stream is a new producer of the caller's buffers: validate intrinsics again
    return TCSFM_OK;
}

int tcsfm_use_own_stream(tcsfm_handle h) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    h->stream = h->own_stream;
    return TCSFM_OK;
}

int tcsfm_synchronize(tcsfm_handle h) {
    if (!h) return TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    if (int rc_ = flush_pending(h)) return rc_;          // queued calls (tcsfm_refine_window_queued) are launched first
    if (int rc_ = join_coalesce_lanes(h)) return rc_;    // (merged sequences that ran on lanes: the handle's stream waits for them)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (tcsfm_ctx *c : h->lanes)
        if (int rc_ = pending_error(c)) { h->err = c->err; return rc_; }
    return pending_error(h);
}

int tcsfm_disp_to_depth(tcsfm_handle h, const tcsfm_opts *o, int64_t n, const float *disp, float *scaled, float *depth) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (!o || !disp || n < 1) return fail(h, TCSFM_E_ARG, "tcsfm_disp_to_depth: bad argument");
    if (!(o->min_depth > 0 && o->max_depth > o->min_depth)) return fail(h, TCSFM_E_ARG, "min_depth/max_depth invalid");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    const float *d_in; float *d_s, *d_d;
    int rc;
    if ((rc = to_dev(h, o, 0, disp, (size_t)n, &d_in))) return rc;
    if ((rc = out_dev(h, o, 1, scaled, (size_t)n, &d_s))) return rc;
    if ((rc = out_dev(h, o, 2, depth, (size_t)n, &d_d))) return rc;
    hipLaunchKernelGGL(k_disp_to_depth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d_in, d_s, d_d, (long long)n,
                       1.f / o->max_depth, 1.f / o->min_depth);
    HIPCHK(h, hipGetLastError());
    if ((rc = copy_back(h, o, scaled, d_s, (size_t)n))) return rc;
    if ((rc = copy_back(h, o, depth, d_d, (size_t)n))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_ssim(tcsfm_handle h, const tcsfm_opts *o, int planes, const float *x, const float *y, float *out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (!o || !x || !y || !out || planes < 1) return fail(h, TCSFM_E_ARG, "tcsfm_ssim: bad argument");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    size_t hw = (size_t)h->H * h->W, n = hw * planes;
    const float *d_x, *d_y; float *d_o;
    int rc;
    if ((rc = to_dev(h, o, 0, x, n, &d_x))) return rc;
    if ((rc = to_dev(h, o, 1, y, n, &d_y))) return rc;
    if ((rc = out_dev(h, o, 2, out, n, &d_o))) return rc;
    hipLaunchKernelGGL(k_ssim, dim3((unsigned)((hw + 255) / 256), planes), dim3(256), 0, h->stream, d_x, d_y, d_o, h->H, h->W);
    HIPCHK(h, hipGetLastError());
    if ((rc = copy_back(h, o, out, d_o, n))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_warp(tcsfm_handle h, const tcsfm_opts *o, int N, const float *src, const float *depth_t, const float *depth_s,
               const float *pose, const float *K, float *img_rec, float *valid, float *proj_depth, float *comp_depth) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!src || !depth_t || !depth_s || !pose || !K) return fail(h, TCSFM_E_ARG, "tcsfm_warp: NULL input");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, N))) return rc;
    size_t hw = (size_t)h->H * h->W;
    const float *d_src, *d_dt, *d_ds, *d_pose, *d_K;
    float *d_rec, *d_valid, *d_pd, *d_cd;
    if ((rc = to_dev(h, o, 0, src, N * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 1, depth_t, N * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 2, depth_s, N * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 3, pose, (size_t)N * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 4, K, (size_t)N * 9, &d_K))) return rc;
    if ((rc = out_dev(h, o, 5, img_rec, N * 3 * hw, &d_rec))) return rc;
    if ((rc = out_dev(h, o, 6, valid, N * hw, &d_valid))) return rc;
    if ((rc = out_dev(h, o, 7, proj_depth, N * hw, &d_pd))) return rc;
    if ((rc = out_dev(h, o, 8, comp_depth, N * hw, &d_cd))) return rc;
    if (o->depth_is_disp) return fail(h, TCSFM_E_ARG, "tcsfm_warp takes depth maps (call tcsfm_disp_to_depth first)");
    if ((rc = run_init(h, o, N, d_pose, nullptr, d_K, 0))) return rc;
    WarpParams P;
    P.src = d_src; P.depth_t = d_dt; P.depth_s = d_ds; P.pc = h->pconst;
    P.rec = d_rec; P.valid = d_valid; P.pd = d_pd; P.cd = d_cd; P.H = h->H; P.W = h->W;
    P.tgt = nullptr; P.posenet_in = nullptr; P.win_B = 0; P.win_S = 0;
    hipLaunchKernelGGL(k_warp, dim3((unsigned)((hw + 255) / 256), N), dim3(256), 0, h->stream, P);
    HIPCHK(h, hipGetLastError());
    if ((rc = copy_back(h, o, img_rec, d_rec, N * 3 * hw))) return rc;
    if ((rc = copy_back(h, o, valid, d_valid, N * hw))) return rc;
    if ((rc = copy_back(h, o, proj_depth, d_pd, N * hw))) return rc;
    if ((rc = copy_back(h, o, comp_depth, d_cd, N * hw))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_warp_posenet_input(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                             const float *depth_s, const float *pose, const float *K, float *posenet_in, float *valid) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!tgt || !src || !depth_t || !depth_s || !pose || !K || !posenet_in) return fail(h, TCSFM_E_ARG, "tcsfm_warp_posenet_input: NULL argument");
    if (o->depth_is_disp) return fail(h, TCSFM_E_ARG, "tcsfm_warp_posenet_input takes depth maps");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, N))) return rc;
    size_t hw = (size_t)h->H * h->W;
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_pose, *d_K;
    float *d_out, *d_valid;
    if ((rc = to_dev(h, o, 0, tgt, N * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, N * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, N * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, N * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, pose, (size_t)N * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 5, K, (size_t)N * 9, &d_K))) return rc;
    if ((rc = out_dev(h, o, 6, posenet_in, N * 6 * hw, &d_out))) return rc;
    if ((rc = out_dev(h, o, 7, valid, N * hw, &d_valid))) return rc;
    if ((rc = run_init(h, o, N, d_pose, nullptr, d_K, 0))) return rc;
    WarpParams P;
    P.src = d_src; P.depth_t = d_dt; P.depth_s = d_ds; P.pc = h->pconst;
    P.rec = nullptr; P.valid = d_valid; P.pd = nullptr; P.cd = nullptr; P.tgt = d_tgt; P.posenet_in = d_out; P.H = h->H; P.W = h->W;
    P.win_B = 0; P.win_S = 0;
    hipLaunchKernelGGL(k_warp, dim3((unsigned)((hw + 255) / 256), N), dim3(256), 0, h->stream, P);
    HIPCHK(h, hipGetLastError());
    if ((rc = copy_back(h, o, posenet_in, d_out, N * 6 * hw))) return rc;
    if ((rc = copy_back(h, o, valid, d_valid, N * hw))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_photometric(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                      const float *depth_s, const float *pose, const float *K, float *diff, float *valid, float *weight,
                      float *auto_err, float *auto_mask, float *img_rec) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!tgt || !src || !depth_t || !depth_s || !pose || !K) return fail(h, TCSFM_E_ARG, "tcsfm_photometric: NULL input");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, N))) return rc;
    size_t hw = (size_t)h->H * h->W;
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_pose, *d_K;
    if ((rc = to_dev(h, o, 0, tgt, N * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, N * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, N * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, N * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, pose, (size_t)N * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 5, K, (size_t)N * 9, &d_K))) return rc;
    float *outs[6] = {diff, valid, weight, auto_err, auto_mask, img_rec}, *d_out[6];
    for (int i = 0; i < 6; i++)
        if ((rc = out_dev(h, o, 6 + i, outs[i], (i == 5 ? 3 : 1) * N * hw, &d_out[i]))) return rc;
    if ((rc = run_pack(h, o, N, d_tgt, d_src, d_dt, d_ds))) return rc;
    if ((rc = run_init(h, o, N, d_pose, nullptr, d_K, 0))) return rc;
    LinParams P = lin_params(h, o, 6);
    P.o_diff = d_out[0]; P.o_valid = d_out[1]; P.o_weight = d_out[2]; P.o_auto_err = d_out[3]; P.o_auto_mask = d_out[4]; P.o_rec = d_out[5];
    launch_lin(h, P, N, 6, false, MODE_MAPS);
    HIPCHK(h, hipGetLastError());
    for (int i = 0; i < 6; i++)
        if ((rc = copy_back(h, o, outs[i], d_out[i], (i == 5 ? 3 : 1) * N * hw))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

// shared by tcsfm_linearize and tcsfm_loss_surface: one evaluation at given poses, results to host doubles
static int eval_once(tcsfm_ctx *h, const tcsfm_opts *o, int N, int Nimg, const float *d_tgt, const float *d_src, const float *d_dt,
                     const float *d_ds, const float *d_pose, const float *d_ls, const float *d_K, int mode, std::vector<double> &host) {
    int rc, np = np_of(o), shared = (Nimg == 1 && N > 1) ? 1 : 0;
    if ((rc = run_pack(h, o, Nimg, d_tgt, d_src, d_dt, d_ds))) return rc;
    if ((rc = run_init(h, o, N, d_pose, d_ls, d_K, shared))) return rc;
    LinParams P = lin_params(h, o, np);
    P.shared_image = shared;
    launch_lin(h, P, N, np, o->w_dc > 0.f, mode);
    HIPCHK(h, hipGetLastError());
    SolveParams S = solve_params(h, o, np, shared);
    S.mode = 2;
    if (mode == MODE_COST) S.has_dc = 0;  // only the three scalar sums are live in cost mode
    launch_solve(h, S, N, np);
    HIPCHK(h, hipGetLastError());
    int rec = np * np + np + 4;
    host.resize((size_t)N * rec);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->lin_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_linearize(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                    const float *depth_s, const float *pose, const float *log_scale, const float *K, double *Hmat, double *g,
                    double *stats) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!tgt || !src || !depth_t || !depth_s || !pose || !K) return fail(h, TCSFM_E_ARG, "tcsfm_linearize: NULL input");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, N))) return rc;
    size_t hw = (size_t)h->H * h->W;
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_pose, *d_K, *d_ls;
    if ((rc = to_dev(h, o, 0, tgt, N * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, N * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, N * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, N * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, pose, (size_t)N * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 5, K, (size_t)N * 9, &d_K))) return rc;
    if ((rc = to_dev(h, o, 6, log_scale, (size_t)N, &d_ls))) return rc;
    std::vector<double> host;
    if ((rc = eval_once(h, o, N, N, d_tgt, d_src, d_dt, d_ds, d_pose, d_ls, d_K, MODE_LIN, host))) return rc;
    int np = np_of(o), rec = np * np + np + 4;
    for (int n = 0; n < N; n++) {
        const double *r = &host[(size_t)n * rec];
        if (Hmat) memcpy(Hmat + (size_t)n * np * np, r, sizeof(double) * np * np);
        if (g) memcpy(g + (size_t)n * np, r + np * np, sizeof(double) * np);
        if (stats) memcpy(stats + (size_t)n * 4, r + np * np + np, sizeof(double) * 4);
    }
    return TCSFM_OK;
}

int tcsfm_linearize_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                           const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *log_scale,
                           double *Hmat, double *g, double *stats) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_linearize_window: need 1 <= 2*B*S <= max_pairs");
    const int N = 2 * B * S;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!tgt || !srcs || !depth_t || !depth_s || !pose || !K) return fail(h, TCSFM_E_ARG, "tcsfm_linearize_window: NULL input");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, B))) return rc;
    const size_t hw = (size_t)h->H * h->W;
    const int np = np_of(o);
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_K, *d_pose, *d_ls;
    if ((rc = to_dev(h, o, 0, tgt, B * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, srcs, (size_t)B * S * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, B * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, (size_t)B * S * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, K, (size_t)B * 9, &d_K))) return rc;
    if ((rc = to_dev(h, o, 5, pose, (size_t)N * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 6, log_scale, (size_t)N, &d_ls))) return rc;
    InitParams I = init_params(h, o, N, d_pose, np == 7 ? d_ls : nullptr, d_K, 0);
    I.K_mod = B;
    if ((rc = run_pack(h, o, N, d_tgt, d_src, d_dt, d_ds, &I, B, S))) return rc;
    LinParams P = lin_params(h, o, np);
    SolveParams Sv = solve_params(h, o, np, 0);
    if (S > 1 && o->argmin) { P.sel_B = B; P.sel_S = S; }
    apply_window_rule(h, o, B, S, N, P, Sv);
    launch_lin(h, P, N, np, o->w_dc > 0.f, MODE_LIN);
    Sv.mode = 2;
    launch_solve(h, Sv, N, np);
    HIPCHK(h, hipGetLastError());
    const int rec = np * np + np + 4;
    std::vector<double> host((size_t)N * rec);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->lin_out, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int n = 0; n < N; n++) {
        const double *r = &host[(size_t)n * rec];
        if (Hmat) memcpy(Hmat + (size_t)n * np * np, r, sizeof(double) * np * np);
        if (g) memcpy(g + (size_t)n * np, r + np * np, sizeof(double) * np);
        if (stats) memcpy(stats + (size_t)n * 4, r + np * np + np, sizeof(double) * 4);
    }
    return TCSFM_OK;
}

int tcsfm_loss_surface(tcsfm_handle h, const tcsfm_opts *o, const float *tgt, const float *src, const float *depth_t,
                       const float *depth_s, const float *K, int P, const float *poses, double *cost_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, P);
    if (rc) return rc;
    if (!tgt || !src || !depth_t || !depth_s || !poses || !K || !cost_out) return fail(h, TCSFM_E_ARG, "tcsfm_loss_surface: NULL argument");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, 1))) return rc;
    size_t hw = (size_t)h->H * h->W;
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_pose, *d_K;
    if ((rc = to_dev(h, o, 0, tgt, 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, poses, (size_t)P * 6, &d_pose))) return rc;
    if ((rc = to_dev(h, o, 5, K, (size_t)9, &d_K))) return rc;
    std::vector<double> host;
    tcsfm_opts oo = *o;
    oo.refine = TCSFM_REFINE_POSE;
    if ((rc = eval_once(h, &oo, P, 1, d_tgt, d_src, d_dt, d_ds, d_pose, nullptr, d_K, MODE_COST, host))) return rc;
    const int rec = 6 * 6 + 6 + 4;
    for (int i = 0; i < P; i++) cost_out[i] = host[(size_t)i * rec + 42];
    return TCSFM_OK;
}

// shared body of tcsfm_refine (win_B == 0: one image set per pair) and tcsfm_refine_window (win_B x win_S window)
// frame-level pack cache of a sequence call (kernels.h k_frame_pack / k_pack_cached): the ring's packed frames, and where this call's windows start
struct FrameCache { const float4 *fpack; const float *fdepth; int slot0, tpos; };

static int refine_body(tcsfm_handle h, const tcsfm_opts *o, int N, int win_B, int win_S, const float *tgt, const float *src,
                       const float *depth_t, const float *depth_s, const float *K, const float *pose_in, const float *log_scale_in,
                       float *pose_out, float *log_scale_out, float *stats_out, const WinOff *wo, const FrameCache *fc,
                       const CoalTab *ct = nullptr, float *const *ct_out = nullptr, float *const *ct_ls_out = nullptr) {
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (ct) {        // coalesced calls (flush_pending): every array argument comes from the table; device pointers, validated at enqueue
        tgt = ct->tgt[0]; src = ct->src[0]; depth_t = ct->dt[0]; depth_s = ct->ds[0]; K = ct->K[0]; pose_in = ct->pose[0]; pose_out = ct_out[0];
    }
    if (!tgt || !src || !depth_t || !depth_s || !pose_in || !pose_out || !K) return fail(h, TCSFM_E_ARG, "tcsfm_refine: NULL input");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    const int nimg_t = ct ? ct->cB : (win_B ? win_B : N), nimg_s = ct ? ct->cB * ct->cS : (win_B ? win_B * win_S : N);   // image sets behind tgt / src
    if (!ct && (rc = check_intrinsics(h, o, K, nimg_t))) return rc;
    const size_t hw = (size_t)h->H * h->W;
    const int np = np_of(o);
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_K, *d_pose_in, *d_ls_in;
    if ((rc = to_dev(h, o, 0, tgt, nimg_t * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, nimg_s * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, nimg_t * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, nimg_s * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, K, (size_t)nimg_t * 9, &d_K))) return rc;
    if ((rc = to_dev(h, o, 5, pose_in, (size_t)N * 6, &d_pose_in))) return rc;
    if ((rc = to_dev(h, o, 6, log_scale_in, (size_t)N, &d_ls_in))) return rc;
    float *d_pose_out, *d_ls_out = nullptr, *d_stats = nullptr;
    if ((rc = out_dev(h, o, 7, pose_out, (size_t)N * 6, &d_pose_out))) return rc;
    if (np == 7 && log_scale_out && (rc = out_dev(h, o, 8, log_scale_out, (size_t)N, &d_ls_out))) return rc;
    const size_t nstats = (size_t)N * (o->n_iters + 1) * TCSFM_NSTAT;
    if (stats_out) {
        if ((rc = out_dev(h, o, 9, stats_out, nstats, &d_stats))) return rc;
        HIPCHK(h, hipMemsetAsync(d_stats, 0, nstats * sizeof(float), h->stream));
    }
    // per-pixel min over the sources: the forward pairs also evaluate the other sources' residuals (k_linearize<SEL>)
    const int n_sel = (win_B && win_S > 1 && o->argmin) ? win_B * win_S : 0;

    InitParams I = init_params(h, o, N, d_pose_in, np == 7 ? d_ls_in : nullptr, d_K, 0);
    I.K_mod = win_B;
    LinParams P = lin_params(h, o, np);
    if (fc) {      // sequence call: rgb + depth were packed once per frame when they landed; only the pair-specific part is formed here
        if (h->tickets_dirty) {
            HIPCHK(h, hipMemsetAsync(h->tickets, 0, (size_t)h->max_pairs * h->ngrp_alloc * sizeof(int), h->stream));
            h->tickets_dirty = false;
        }
        PackCachedParams C;
        C.fpack = fc->fpack; C.tgtpack = h->tgtpack; C.pair_src = h->pair_idx; C.pair_dep = h->pair_idx + h->max_pairs;
        C.H = h->H; C.W = h->W; C.N = N; C.win_B = win_B; C.win_S = win_S; C.slot0 = fc->slot0; C.tpos = fc->tpos; C.win_off = *wo;
        C.wl = o->w_l1 / 3.f; C.ws = o->w_ssim / 3.f; C.init = I;
        {
            ProfScope prof(h, 2);
            hipLaunchKernelGGL(k_pack_cached, dim3((unsigned)((hw + 255) / 256), N / 2), dim3(256), 0, h->stream, C);      // (a row per forward pair)
        }
        HIPCHK(h, hipGetLastError());
        P.srcpack = fc->fpack; P.depth_t = fc->fdepth; P.pair_src = C.pair_src; P.pair_dep = C.pair_dep;
    } else {
        // window forms: every image packed ONCE (LinParams::tshare, kernels.h) -- TCSFM_TSHARE=0 keeps the round-4 layout (A/B hook)
        static const bool tshare_env = !(getenv("TCSFM_TSHARE") && atoi(getenv("TCSFM_TSHARE")) == 0);
        const bool tshare = tshare_env && (win_B > 0 || ct != nullptr);
        if ((rc = run_pack(h, o, N, d_tgt, d_src, d_dt, d_ds, &I, win_B, win_S, nullptr, wo, ct, nullptr, nullptr, 0, nullptr, tshare))) return rc;
        if (tshare) { P.tshare = 1; P.tshare_sb = N / 2; }
    }
    SolveParams S = solve_params(h, o, np, 0);
    S.stats = d_stats;
    if (ct) {
        S.c_ncall = ct->ncall; S.c_B = ct->cB; S.c_S = ct->cS;
        for (int i = 0; i < ct->ncall; i++) { S.c_pose_out[i] = ct_out[i]; S.c_ls_out[i] = ct_ls_out ? ct_ls_out[i] : nullptr; }
    }
    const bool dc = o->w_dc > 0.f;
    const bool lm = o->solver == TCSFM_SOLVER_LM;
    if (n_sel) { P.sel_B = win_B; P.sel_S = win_S; }   // min over the sources: evaluated inside k_linearize<SEL>
    apply_window_rule(h, o, win_B, win_S, N, P, S);
    if ((rc = trace_check(h, o, N))) return rc;
    for (int it = 0; it < o->n_iters; it++) {
        trace_at(h, it, N, P, S);
        launch_lin(h, P, N, np, dc, MODE_LIN);
        S.it = it; S.mode = 0;
        const bool last = !lm && it == o->n_iters - 1;   // the last solve also emits the refined pose
        S.pose_out = last ? d_pose_out : nullptr; S.log_scale_out = last ? d_ls_out : nullptr;
        launch_solve(h, S, N, np);
    }
    if (lm && o->n_iters > 0) {  // cost-only pass deciding whether the last step is kept
        trace_at(h, o->n_iters, N, P, S);
        launch_lin(h, P, N, np, dc, MODE_COST);
        S.it = o->n_iters; S.mode = 1;
        S.pose_out = d_pose_out; S.log_scale_out = d_ls_out;
        launch_solve(h, S, N, np);
    }
    HIPCHK(h, hipGetLastError());
    if (o->n_iters == 0) {  // nothing to solve: round-trip the pose through SE(3)
        FinishParams F;
        F.st = h->state; F.pose_out = d_pose_out; F.log_scale_out = d_ls_out; F.N = N;
        hipLaunchKernelGGL(k_finish, dim3((N + 63) / 64), dim3(64), 0, h->stream, F);
        HIPCHK(h, hipGetLastError());
    }
    if ((rc = copy_back(h, o, pose_out, d_pose_out, (size_t)N * 6))) return rc;
    if (d_ls_out && (rc = copy_back(h, o, log_scale_out, d_ls_out, (size_t)N))) return rc;
    if ((rc = copy_back(h, o, stats_out, d_stats, nstats))) return rc;
    if (o->host_ptrs == 1) HIPCHK(h, hipStreamSynchronize(h->stream));   // host_ptrs == 2: pinned + asynchronous, the caller synchronises
    return TCSFM_OK;
}

// Graph replay (tcsfm_set_graph_replay).  A B = 1 refinement is nine short launches: ~42 us of host time against ~40 us of GPU time
// per call with three calls in flight -- on a slow host the launches, not the kernels, set the rate.  A call whose arguments (options,
// sizes, every pointer) equal those of an earlier call on this handle / lane is captured once (the second time it is seen: the first
// run validates the intrinsics and allocates scratch, which a capture must not) and replayed with ONE hipGraphLaunch from then on.
// The kernels, their order and their arguments are exactly those of the plain path (`body`): results are bit-identical.
// kind: 0 pose refinement, 1 dense refinement (the joint dense mode forks its inverse pairs onto a second stream and joins it again
// through events: that fork / join is captured with it).
extern "C++" {
template <class Body>
static int replay_or_run(tcsfm_ctx *h, const tcsfm_opts *o, int kind, int N, int win_B, int win_S, const void *const *ptrs, const float *K,
                         bool bypass, Body body) {
    const bool eligible = h && o && h->graph_slots > 0 && !h->capturing && !o->host_ptrs && !bypass && !h->profiling && !h->trace_bits &&
                          !h->trace_decide && !h->dbg_stamps && h->stream != nullptr;
    if (!eligible) return body();
    tcsfm_ctx::CallKey key;
    memset(&key, 0, sizeof(key));
    key.o = *o; key.N = N; key.win_B = win_B; key.win_S = win_S; key.pad = kind;
    memcpy(key.p, ptrs, sizeof(key.p));
    tcsfm_ctx::CallGraph *e = nullptr;
    for (auto &g : h->graphs) if (!memcmp(&g.key, &key, sizeof(key))) { e = &g; break; }
    if (e && e->exec) {                                    // replay
        int rc = check_common(h, o, N);
        if (rc) return rc;
        DeviceGuard dev_guard(h->device);
        if (int rc_ = pending_error(h)) return rc_;
        e->used = ++h->graph_clock;
        if (h->tickets_dirty) {                            // (a failed call may have left group tickets non-zero: what run_pack would do)
            HIPCHK(h, hipMemsetAsync(h->tickets, 0, (size_t)h->max_pairs * h->ngrp_alloc * sizeof(int), h->stream));
            h->tickets_dirty = false;
        }
        if (kind == 1) { if (int rc_ = dref_clean(h)) return rc_; }      // (what dense_ref_run does before its first launch)
        HIPCHK(h, hipGraphLaunch(e->exec, h->stream));
        h->graph_replays++;
        return TCSFM_OK;
    }
    if (!e) {                                              // first sighting: plain run, remember the call
        if ((int)h->graphs.size() >= h->graph_slots) {     // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); i++) if (h->graphs[i].used < h->graphs[lru].used) lru = i;
            if (h->graphs[lru].exec) { sync_graph_streams(h); (void)hipGraphExecDestroy(h->graphs[lru].exec); }      // (it may still be running)
            h->graphs.erase(h->graphs.begin() + lru);
        }
        tcsfm_ctx::CallGraph g;
        g.key = key; g.exec = nullptr; g.seen = 1; g.used = ++h->graph_clock;
        h->graphs.push_back(g);
        return body();
    }
    e->used = ++h->graph_clock;
    const int nimg_t = win_B ? win_B : N;
    if (e->seen != 1 || !k_known(h, K, nimg_t))           // marked uncapturable, or the intrinsics would be re-validated (a blocking copy)
        return body();
    {                                                      // second sighting: capture the plain path's launches
        DeviceGuard dev_guard(h->device);
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            e->seen = -1;
            return body();
        }
        h->capturing = true;
        const std::string err_before = h->err;
        int rc = body();
        h->capturing = false;
        hipGraph_t graph = nullptr;
        hipError_t ce = hipStreamEndCapture(h->stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc == TCSFM_OK && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            (void)hipGraphDestroy(graph);
            e->exec = exec;
            h->graph_captures++;
            HIPCHK(h, hipGraphLaunch(exec, h->stream));
            return TCSFM_OK;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        e->seen = -1;                                      // not capturable in this configuration: plain launches from now on
        h->err = err_before;
    }
    return body();
}
}  // extern "C++"

static int refine_impl(tcsfm_handle h, const tcsfm_opts *o, int N, int win_B, int win_S, const float *tgt, const float *src,
                       const float *depth_t, const float *depth_s, const float *K, const float *pose_in, const float *log_scale_in,
                       float *pose_out, float *log_scale_out, float *stats_out, const WinOff *wo = nullptr, const FrameCache *fc = nullptr) {
    const void *ptrs[10] = {tgt, src, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out};
    return replay_or_run(h, o, 0, N, win_B, win_S, ptrs, K, wo != nullptr || fc != nullptr, [&]() {
        return refine_body(h, o, N, win_B, win_S, tgt, src, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out, wo, fc);
    });
}

int tcsfm_refine(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                 const float *depth_s, const float *K, const float *pose_in, const float *log_scale_in, float *pose_out,
                 float *log_scale_out, float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    return refine_impl(h, o, N, 0, 0, tgt, src, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out);
}

int tcsfm_refine_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                        const float *depth_t, const float *depth_s, const float *K, const float *pose_in,
                        const float *log_scale_in, float *pose_out, float *log_scale_out, float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_window: need 1 <= 2*B*S <= max_pairs");
    return refine_impl(h, o, 2 * B * S, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out);
}

int tcsfm_scale_recovery(tcsfm_handle h, const tcsfm_opts *o, int N, const float *depth, const float *K, float real_cam_height,
                         int pad_to_batch, float *scale_out, float *median_out, float *height_out, float *mask_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!depth || !K || !scale_out) return fail(h, TCSFM_E_ARG, "tcsfm_scale_recovery: NULL argument");
    if (h->H < 5 || h->W < 5) return fail(h, TCSFM_E_ARG, "tcsfm_scale_recovery: image too small");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, N))) return rc;
    const size_t hw = (size_t)h->H * h->W;
    if (!h->scale_keys) {
        HIPCHK(h, hipMalloc((void **)&h->scale_keys, (size_t)h->max_pairs * hw * sizeof(unsigned)));
        HIPCHK(h, hipMalloc((void **)&h->scale_hist, 260 * sizeof(unsigned)));
    }
    const float *d_depth, *d_K;
    float *d_scale, *d_med, *d_h, *d_m;
    if ((rc = to_dev(h, o, 0, depth, N * hw, &d_depth))) return rc;
    if ((rc = to_dev(h, o, 1, K, (size_t)N * 9, &d_K))) return rc;
    if ((rc = out_dev(h, o, 2, scale_out, (size_t)1, &d_scale))) return rc;
    if ((rc = out_dev(h, o, 3, median_out, (size_t)1, &d_med))) return rc;
    if ((rc = out_dev(h, o, 4, height_out, N * hw, &d_h))) return rc;
    if ((rc = out_dev(h, o, 5, mask_out, N * hw, &d_m))) return rc;
    HIPCHK(h, hipMemsetAsync(h->scale_hist, 0, 260 * sizeof(unsigned), h->stream));
    GroundParams G;
    G.depth = d_depth; G.K = d_K; G.height = d_h; G.mask = d_m; G.keys = h->scale_keys; G.H = h->H; G.W = h->W; G.err = h->err_dev;
    hipLaunchKernelGGL(k_ground, dim3((unsigned)((hw + 255) / 256), N), dim3(256), 0, h->stream, G);
    // the reference pads the batch to config['minibatch'] with copies of image 0 (dnet_layers.py:307-311): weight image 0
    const int w0 = 1 + (pad_to_batch > N ? pad_to_batch - N : 0);
    unsigned *hist = h->scale_hist, *state = h->scale_hist + 256;
    const int nb = (int)((hw + 255) / 256) < 64 ? (int)((hw + 255) / 256) : 64;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_sel_hist, dim3(nb, N), dim3(256), 0, h->stream, (const unsigned *)h->scale_keys, (int)hw, w0,
                           (const unsigned *)state, shift, hist);
        hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(64), 0, h->stream, hist, state, shift, real_cam_height, d_scale, d_med);
    }
    HIPCHK(h, hipGetLastError());
    if ((rc = copy_back(h, o, scale_out, d_scale, (size_t)1))) return rc;
    if ((rc = copy_back(h, o, median_out, d_med, (size_t)1))) return rc;
    if ((rc = copy_back(h, o, height_out, d_h, N * hw))) return rc;
    if ((rc = copy_back(h, o, mask_out, d_m, N * hw))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_smooth_loss(tcsfm_handle h, const tcsfm_opts *o, int N, const float *disp, const float *img, double *loss_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!disp || !img || !loss_out) return fail(h, TCSFM_E_ARG, "tcsfm_smooth_loss: NULL argument");
    if (h->H < 2 || h->W < 2) return fail(h, TCSFM_E_ARG, "tcsfm_smooth_loss: image too small");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    const size_t hw = (size_t)h->H * h->W;
    const int nb = (int)((hw + 255) / 256);
    const float *d_disp, *d_img;
    if ((rc = to_dev(h, o, 0, disp, N * hw, &d_disp))) return rc;
    if ((rc = to_dev(h, o, 1, img, N * 3 * hw, &d_img))) return rc;
    // scratch: N means (double) + N * nb * 2 partial sums, carved from the staging slot 2
    float *scratch;
    tcsfm_opts os = *o; os.host_ptrs = 1;
    const size_t nfl = (size_t)N * 2 + (size_t)N * nb * 2;
    float dummy;
    if ((rc = out_dev(h, &os, 2, &dummy, nfl, &scratch))) return rc;
    double *mean = reinterpret_cast<double *>(scratch);
    float *partial = scratch + (size_t)N * 2;
    hipLaunchKernelGGL(k_smooth_mean, dim3(N), dim3(1024), 0, h->stream, d_disp, (int)hw, mean);
    hipLaunchKernelGGL(k_smooth, dim3(nb, N), dim3(256), 0, h->stream, d_disp, d_img, (const double *)mean, h->H, h->W, partial);
    HIPCHK(h, hipGetLastError());
    std::vector<float> hp((size_t)N * nb * 2);
    HIPCHK(h, hipMemcpyAsync(hp.data(), partial, hp.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double sx = 0.0, sy = 0.0;
    for (size_t i = 0; i < hp.size(); i += 2) { sx += hp[i]; sy += hp[i + 1]; }
    *loss_out = sx / ((double)N * h->H * (h->W - 1)) + sy / ((double)N * (h->H - 1) * h->W);
    return TCSFM_OK;
}

// shared body of tcsfm_refine_dense (win_B == 0) and tcsfm_refine_dense_window
static int dense_body(tcsfm_handle h, const tcsfm_opts *o, int N, int win_B, int win_S, const float *tgt, const float *src,
                      const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                      float *depth_out, float *stats_out, const WinOff *wo, const CoalTab *ct = nullptr, float *const *ct_pose = nullptr,
                      float *const *ct_depth = nullptr) {
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (ct) {        // coalesced dense calls (flush_pending): per-pair Gauss-Newton form only, every array argument comes from the table
        tgt = ct->tgt[0]; src = ct->src[0]; depth_t = ct->dt[0]; depth_s = ct->ds[0]; K = ct->K[0]; pose_in = ct->pose[0];
        pose_out = ct_pose[0]; depth_out = ct_depth[0];
    }
    if (!tgt || !src || !depth_t || !depth_s || !pose_in || !pose_out || !depth_out || !K) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: NULL argument");
    const bool ref_mode = win_B && o->window_rule == TCSFM_WINDOW_REFERENCE;      // the reference's own loss (dense_ref_kernel.h)
    if (o->w_dc > 0.f && !ref_mode) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: w_dc must be 0 (use prior_depth), except under window_rule = TCSFM_WINDOW_REFERENCE");
    if (ref_mode && (win_S > JMAXS || o->solver != TCSFM_SOLVER_GN || !(o->prior_init >= 0.f)))
        return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window: window_rule REFERENCE needs S <= 3, the Gauss-Newton solver and prior_init >= 0");
    if (!(o->w_pose_consist >= 0.f) || (o->w_pose_consist > 0.f && (!ref_mode || o->param != TCSFM_PARAM_SE3)))
        return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: w_pose_consist needs the window form under window_rule = TCSFM_WINDOW_REFERENCE and the SE(3) chart");
    if (o->free_source_depths != 0 && !ref_mode)
        return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: free_source_depths needs the window form under window_rule = TCSFM_WINDOW_REFERENCE");
    if (!(o->w_smooth >= 0.f) || (o->w_smooth > 0.f && !ref_mode))
        return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: w_smooth needs the window form under window_rule = TCSFM_WINDOW_REFERENCE");
    if (o->depth_param != TCSFM_DEPTH_FULL && !(o->depth_param == TCSFM_DEPTH_QUARTER && ref_mode && h->H % 4 == 0 && h->W % 4 == 0))
        return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: depth_param QUARTER needs window_rule = TCSFM_WINDOW_REFERENCE (window form) and H, W multiples of 4");
    if (o->param != TCSFM_PARAM_SE3) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense: SE(3) chart only");
    if (!(o->min_depth > 0 && o->max_depth > o->min_depth)) return fail(h, TCSFM_E_ARG, "min_depth/max_depth invalid");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    const int nimg_t = ct ? ct->cB : (win_B ? win_B : N), nimg_s = ct ? ct->cB * ct->cS : (win_B ? win_B * win_S : N);   // image sets behind tgt / src
    if (!ct && (rc = check_intrinsics(h, o, K, nimg_t))) return rc;
    const size_t hw = (size_t)h->H * h->W, n = h->max_pairs;
    const int n_sel = (win_B && win_S > 1 && o->argmin) ? win_B * win_S : 0;
    if (ct && !ref_mode && (n_sel || o->solver != TCSFM_SOLVER_GN || o->host_ptrs || o->n_iters < 1 || (o->dense_joint && win_S >= 2)))
        return fail(h, TCSFM_E_ARG, "internal: only per-pair Gauss-Newton dense calls on device pointers are merged");
    if (ct && ref_mode && (o->host_ptrs || o->n_iters < 1 || stats_out)) return fail(h, TCSFM_E_ARG, "internal: merged reference-loss calls take device pointers and no statistics");
    if (n_sel && !h->sel_maps) HIPCHK(h, hipMalloc((void **)&h->sel_maps, (size_t)2 * h->max_pairs * hw * sizeof(float)));
    const bool lm = o->solver == TCSFM_SOLVER_LM;
    if (lm && !h->dense_rec_acc) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec_acc, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth_acc, n * hw * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->lm_accept, n * sizeof(int)));
    }
    if (!h->dense_rec) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth0, n * hw * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->delta, n * 8 * sizeof(double)));
    }
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_K, *d_pose_in;
    if ((rc = to_dev(h, o, 0, tgt, nimg_t * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, src, nimg_s * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, nimg_t * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, nimg_s * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, K, (size_t)nimg_t * 9, &d_K))) return rc;
    if ((rc = to_dev(h, o, 5, pose_in, (size_t)N * 6, &d_pose_in))) return rc;
    float *d_pose_out, *d_depth_out, *d_stats = nullptr;
    if ((rc = out_dev(h, o, 7, pose_out, (size_t)N * 6, &d_pose_out))) return rc;
    if ((rc = out_dev(h, o, 8, depth_out, N * hw, &d_depth_out))) return rc;
    const size_t nstats = (size_t)N * (o->n_iters + 1) * TCSFM_NSTAT;
    if (stats_out) {
        if ((rc = out_dev(h, o, 9, stats_out, nstats, &d_stats))) return rc;
        HIPCHK(h, hipMemsetAsync(d_stats, 0, nstats * sizeof(float), h->stream));
    }
    if (ref_mode) {
        rc = win_S == 1 ? dense_ref_run<1>(h, o, win_B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose_in, d_pose_out, d_depth_out, d_stats, wo, nullptr, ct, ct_pose, ct_depth)
           : win_S == 2 ? dense_ref_run<2>(h, o, win_B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose_in, d_pose_out, d_depth_out, d_stats, wo, nullptr, ct, ct_pose, ct_depth)
                        : dense_ref_run<3>(h, o, win_B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose_in, d_pose_out, d_depth_out, d_stats, wo, nullptr, ct, ct_pose, ct_depth);
        if (rc) return rc;
        if ((rc = copy_back(h, o, pose_out, d_pose_out, (size_t)N * 6))) return rc;
        if ((rc = copy_back(h, o, depth_out, d_depth_out, N * hw))) return rc;
        if ((rc = copy_back(h, o, stats_out, d_stats, nstats))) return rc;
        if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
        return TCSFM_OK;
    }
    if (win_B && o->dense_joint && win_S >= 2 && win_S <= JMAXS) {   // one depth map per target, 6S x 6S reduced system (joint_kernel.h)
        rc = win_S == 2 ? dense_joint_run<2>(h, o, win_B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose_in, d_pose_out, d_depth_out, d_stats, wo)
                        : dense_joint_run<3>(h, o, win_B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose_in, d_pose_out, d_depth_out, d_stats, wo);
        if (rc) return rc;
        if ((rc = copy_back(h, o, pose_out, d_pose_out, (size_t)N * 6))) return rc;
        if ((rc = copy_back(h, o, depth_out, d_depth_out, N * hw))) return rc;
        if ((rc = copy_back(h, o, stats_out, d_stats, nstats))) return rc;
        if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
        return TCSFM_OK;
    }
    tcsfm_opts oo = *o;
    oo.refine = TCSFM_REFINE_POSE;
    oo.window_rule = TCSFM_WINDOW_PAIR;
    InitParams I = init_params(h, &oo, N, d_pose_in, nullptr, d_K, 0);
    I.K_mod = win_B;
    // Gauss-Newton in the pair form: the back-substitution of iteration k is fused into the linearisation of iteration k+1 (depth
    // maps and per-pixel records ping-pong between two buffers); LM rolls maps back and the min over sources needs the current map
    // materialised before the linearisation, so those keep the separate update launch
    const bool fuse = !lm && !n_sel && o->n_iters > 0;
    if (fuse && !h->dense_rec2) {
        HIPCHK(h, hipMalloc((void **)&h->dense_rec2, n * hw * 8 * sizeof(float)));
        HIPCHK(h, hipMalloc((void **)&h->depth_alt, n * hw * sizeof(float)));
    }
    // every pair gets its OWN copy of its target's depth; the pack also leaves the prior centre depth0
    if ((rc = run_pack(h, &oo, N, d_tgt, d_src, d_dt, d_ds, &I, win_B, win_S, h->depth0, wo, ct))) return rc;
    // the dense kernel's own tile grid (32x16 tiles of 512 threads, as k_linearize; 16x16 / 256 threads measured slower except
    // for 320x240 at B=1: 10.9 vs 11.8 us per launch there, 234 vs 200 us with the chip full) and reduction-group count
    constexpr int DTW = 32, DTH = 16, DNT = 512;
    LinParams P = lin_params(h, &oo, 6);
    P.tiles_x = (h->W + DTW - 1) / DTW; P.tiles_y = (h->H + DTH - 1) / DTH;
    const int nblk = P.tiles_x * P.tiles_y;
    P.ngrp = (nblk + RG - 1) / RG;
    if ((size_t)nblk > (size_t)h->nblk_alloc || P.ngrp > h->ngrp_alloc) return fail(h, TCSFM_E_ARG, "internal: dense tile grid exceeds scratch");
    SolveParams S = solve_params(h, &oo, 6, 0);
    P.direct = nblk <= 512;
    S.partials = P.direct ? h->blockrec : h->partials; S.ngrp = P.direct ? nblk : P.ngrp;
    S.stats = d_stats; S.delta_out = h->delta;
    if (ct) {
        S.c_ncall = ct->ncall; S.c_B = ct->cB; S.c_S = ct->cS;
        for (int i = 0; i < ct->ncall; i++) S.c_pose_out[i] = ct_pose[i];
    }
    DenseParams Dn;
    Dn.dense_rec = h->dense_rec; Dn.depth0 = h->depth0; Dn.lambda_depth = o->lambda_depth; Dn.w_prior = o->prior_depth;
    Dn.prev_rec = nullptr; Dn.prev_delta = h->delta; Dn.depth_next = nullptr;
    Dn.rho_lo = 1.f / o->max_depth; Dn.rho_hi = 1.f / o->min_depth;
    DenseUpdateParams U;
    memset(&U, 0, sizeof(U));
    U.dense_rec = h->dense_rec; U.delta = h->delta; U.depth = h->depth_work; U.depth_out = h->depth_work; U.hw = (int)hw;
    U.rho_lo = Dn.rho_lo; U.rho_hi = Dn.rho_hi;
    float *Dbuf[2] = {h->depth_work, h->depth_alt}, *Rbuf[2] = {h->dense_rec, h->dense_rec2};
    // min over the sources (window form): selection masks of the forward pairs from their residual maps at the current poses
    // AND current depth copies, rebuilt before every linearisation (same two launches as in the pose mode)
    float *sel_diff = h->sel_maps, *sel_valid = h->sel_maps ? h->sel_maps + (size_t)h->max_pairs * hw : nullptr;
    if (n_sel) { P.ext_diff = sel_diff; P.ext_valid = sel_valid; P.n_ext = n_sel; P.ext_B = win_B; P.ext_S = win_S; }
    auto select_pass = [&]() {
        LinParams M = lin_params(h, &oo, 6);
        M.o_diff = sel_diff; M.o_valid = sel_valid;
        launch_lin(h, M, n_sel, 6, false, MODE_MAPS, 2);          // (the selection itself: ext_selected, inside the dense kernel)
    };
    auto linearize = [&]() {
        if (n_sel) select_pass();
        take_stamp(h, P, (size_t)nblk * N);
        ProfScope prof(h, 0);
        if (P.trace != nullptr) hipLaunchKernelGGL((k_dense_linearize<DTW, DTH, DNT, true>), dim3(nblk, N), dim3(DNT), 0, h->stream, P, Dn);
        else hipLaunchKernelGGL((k_dense_linearize<DTW, DTH, DNT>), dim3(nblk, N), dim3(DNT), 0, h->stream, P, Dn);
    };
    DenseLmParams Ul;
    Ul.rec_try = h->dense_rec; Ul.rec_acc = h->dense_rec_acc; Ul.depth_acc = h->depth_acc; Ul.depth = h->depth_work; Ul.delta = h->delta;
    Ul.accept = h->lm_accept; Ul.hw = (int)hw; Ul.rho_lo = U.rho_lo; Ul.rho_hi = U.rho_hi;
    S.accept_out = lm ? h->lm_accept : nullptr;
    const dim3 px_grid((unsigned)((hw + 255) / 256), N);
    if ((rc = trace_check(h, o, N))) return rc;
    for (int it = 0; it < o->n_iters; it++) {
        trace_at(h, it, N, P, S);
        if (fuse) {   // reads depth_{it-1} + the records / step of iteration it-1, leaves depth_it and the records of iteration it
            P.depth_t = Dbuf[it & 1]; Dn.depth_next = Dbuf[(it + 1) & 1];
            Dn.dense_rec = Rbuf[it & 1]; Dn.prev_rec = it > 0 ? Rbuf[(it - 1) & 1] : nullptr;
        }
        linearize();
        S.it = it; S.mode = 0;
        const bool last = !lm && it == o->n_iters - 1;
        S.pose_out = last ? d_pose_out : nullptr; S.log_scale_out = nullptr;
        launch_solve(h, S, N, 6);
        if (lm) hipLaunchKernelGGL(k_dense_update_lm, px_grid, dim3(256), 0, h->stream, Ul);
        else if (!fuse) hipLaunchKernelGGL(k_dense_update, px_grid, dim3(256), 0, h->stream, U);
    }
    if (fuse) {   // the last back-substitution writes the caller's depth map directly
        const int nit = o->n_iters;
        U.dense_rec = Rbuf[(nit - 1) & 1]; U.depth = Dbuf[nit & 1]; U.depth_out = d_depth_out;
        if (ct) {
            U.c_ncall = ct->ncall; U.c_B = ct->cB; U.c_S = ct->cS;
            for (int i = 0; i < ct->ncall; i++) U.c_depth_out[i] = ct_depth[i];
        }
        hipLaunchKernelGGL(k_dense_update, px_grid, dim3(256), 0, h->stream, U);
    }
    if (lm && o->n_iters > 0) {   // evaluate the last trial once more; keep it only if it lowered the cost (pose and depth map)
        trace_at(h, o->n_iters, N, P, S);
        linearize();
        S.it = o->n_iters; S.mode = 1;
        S.pose_out = d_pose_out; S.log_scale_out = nullptr;
        launch_solve(h, S, N, 6);
        hipLaunchKernelGGL(k_dense_final_lm, px_grid, dim3(256), 0, h->stream, (const int *)h->lm_accept, (const float *)h->depth_acc, h->depth_work, (int)hw);
    }
    HIPCHK(h, hipGetLastError());
    if (o->n_iters == 0) {
        FinishParams F;
        F.st = h->state; F.pose_out = d_pose_out; F.log_scale_out = nullptr; F.N = N;
        hipLaunchKernelGGL(k_finish, dim3((N + 63) / 64), dim3(64), 0, h->stream, F);
    }
    if (!fuse) HIPCHK(h, hipMemcpyAsync(d_depth_out, h->depth_work, N * hw * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if ((rc = copy_back(h, o, pose_out, d_pose_out, (size_t)N * 6))) return rc;
    if ((rc = copy_back(h, o, depth_out, d_depth_out, N * hw))) return rc;
    if ((rc = copy_back(h, o, stats_out, d_stats, nstats))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

static int dense_impl(tcsfm_handle h, const tcsfm_opts *o, int N, int win_B, int win_S, const float *tgt, const float *src,
                      const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                      float *depth_out, float *stats_out, const WinOff *wo = nullptr) {
    const void *ptrs[10] = {tgt, src, depth_t, depth_s, K, pose_in, nullptr, pose_out, depth_out, stats_out};
    return replay_or_run(h, o, 1, N, win_B, win_S, ptrs, K, wo != nullptr, [&]() {
        return dense_body(h, o, N, win_B, win_S, tgt, src, depth_t, depth_s, K, pose_in, pose_out, depth_out, stats_out, wo);
    });
}

int tcsfm_refine_dense(tcsfm_handle h, const tcsfm_opts *o, int N, const float *tgt, const float *src, const float *depth_t,
                       const float *depth_s, const float *K, const float *pose_in, float *pose_out, float *depth_out,
                       float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    return dense_impl(h, o, N, 0, 0, tgt, src, depth_t, depth_s, K, pose_in, pose_out, depth_out, stats_out);
}

int tcsfm_refine_dense_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                              const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                              float *depth_out, float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window: need 1 <= 2*B*S <= max_pairs");
    return dense_impl(h, o, 2 * B * S, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, depth_out, stats_out);
}

// ---- coalesced calls: queued B-window calls of one shape as ONE launch sequence (include/tcsfm.h)
// the handle's stream is ordered behind the merged sequences that ran on lanes (consumers queued on it see their poses)
static int join_coalesce_lanes(tcsfm_ctx *h) {
    for (int l = 1; l <= (int)h->lanes.size() && h->coal_dirty; l++)
        if (h->coal_dirty & (1u << l)) {
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->lanes[l - 1]->done_ev, 0));
            h->coal_dirty &= ~(1u << l);
        }
    return TCSFM_OK;
}

static int flush_pending(tcsfm_ctx *h) {
    if (h->pending.empty()) return TCSFM_OK;
    std::vector<tcsfm_ctx::PendingCall> calls;
    calls.swap(h->pending);
    const tcsfm_opts o = h->pend_opts;
    const int B = h->pend_B, S = h->pend_S, n = (int)calls.size();
    // merged sequences alternate over coal_lanes streams (tcsfm_set_coalesce_lanes): sequence k runs on lane k mod coal_lanes, behind
    // everything queued on the handle's stream so far; the short tail kernels of one sequence (k_solve: 20 workgroups, 6 us) then
    // overlap the other's chip-filling launches.  tcsfm_flush / tcsfm_synchronize order the handle's stream behind them.
    const int nl = h->lanes_serial ? 1 : std::min(h->coal_lanes, (int)h->lanes.size() + 1), l = nl > 1 ? h->coal_batches % nl : 0;
    tcsfm_ctx *c = l == 0 ? h : h->lanes[l - 1];
    h->coal_batches++; h->coal_calls += n;
    if (c != h) {
        HIPCHK(h, hipEventRecord(c->in_ev, h->stream));
        HIPCHK(h, hipStreamWaitEvent(c->own_stream, c->in_ev, 0));
        c->stream = c->own_stream;
    }
    int rc;
    const bool dense = h->pend_dense != 0;
    if (n == 1) {
        const auto &q = calls[0];
        rc = dense ? dense_impl(c, &o, 2 * B * S, B, S, q.tgt, q.srcs, q.dt, q.ds, q.K, q.pose_in, q.pose_out, q.depth_out, nullptr)
                   : refine_impl(c, &o, 2 * B * S, B, S, q.tgt, q.srcs, q.dt, q.ds, q.K, q.pose_in, q.ls_in, q.pose_out, q.ls_out, nullptr);
    } else if (dense) {
        CoalTab ct;
        memset(&ct, 0, sizeof(ct));
        ct.ncall = n; ct.cB = B; ct.cS = S;
        float *outs[TC_MAX_COAL], *douts[TC_MAX_COAL];
        for (int i = 0; i < n; i++) {
            ct.tgt[i] = calls[i].tgt; ct.src[i] = calls[i].srcs; ct.dt[i] = calls[i].dt; ct.ds[i] = calls[i].ds; ct.K[i] = calls[i].K; ct.pose[i] = calls[i].pose_in;
            outs[i] = calls[i].pose_out; douts[i] = calls[i].depth_out;
        }
        rc = dense_body(c, &o, 2 * B * S * n, B * n, S, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ct, outs, douts);
    } else {
        CoalTab ct;
        memset(&ct, 0, sizeof(ct));
        ct.ncall = n; ct.cB = B; ct.cS = S;
        float *outs[TC_MAX_COAL];
        for (int i = 0; i < n; i++) {
            ct.tgt[i] = calls[i].tgt; ct.src[i] = calls[i].srcs; ct.dt[i] = calls[i].dt; ct.ds[i] = calls[i].ds; ct.K[i] = calls[i].K; ct.pose[i] = calls[i].pose_in;
            outs[i] = calls[i].pose_out;
        }
        float *louts[TC_MAX_COAL];
        for (int i = 0; i < n; i++) { ct.ls[i] = calls[i].ls_in; louts[i] = calls[i].ls_out; }
        rc = refine_body(c, &o, 2 * B * S * n, B * n, S, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ct, outs, louts);
    }
    if (c != h) {
        if (rc) { h->err = c->err; return rc; }
        HIPCHK(h, hipEventRecord(c->done_ev, c->own_stream));
        h->coal_dirty |= 1u << l;
    }
    return rc;
}

int tcsfm_set_coalesce(tcsfm_handle h, int max_calls) {
    if (!h) return TCSFM_E_ARG;
    if (max_calls < 0 || max_calls > TC_MAX_COAL) return fail(h, TCSFM_E_ARG, "tcsfm_set_coalesce: 0 <= max_calls <= 16");
    int rc = flush_pending(h);
    h->coal_max = max_calls;
    return rc;
}

int tcsfm_set_coalesce_lanes(tcsfm_handle h, int n_streams) {
    if (!h) return TCSFM_E_ARG;
    if (n_streams < 1 || n_streams > (int)h->lanes.size() + 1) return fail(h, TCSFM_E_ARG, "tcsfm_set_coalesce_lanes: 1 <= n_streams <= lanes of the handle (tcsfm_set_lanes)");
    DeviceGuard dev_guard(h->device);
    int rc = flush_pending(h);
    if (!rc) rc = join_coalesce_lanes(h);
    h->coal_lanes = n_streams;
    return rc;
}

int tcsfm_flush(tcsfm_handle h) {
    if (!h) return TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    int rc = flush_pending(h);
    return rc ? rc : join_coalesce_lanes(h);
}

int tcsfm_coalesce_counts(tcsfm_handle h, int *batches, int *calls) {
    if (!h) return TCSFM_E_ARG;
    if (batches) *batches = h->coal_batches;
    if (calls) *calls = h->coal_calls;
    return TCSFM_OK;
}

int tcsfm_refine_window_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs, const float *depth_t,
                               const float *depth_s, const float *K, const float *pose_in, float *pose_out) {
    return tcsfm_refine_window_scale_queued(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, nullptr, pose_out, nullptr);
}

int tcsfm_refine_window_scale_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs, const float *depth_t,
                                     const float *depth_s, const float *K, const float *pose_in, const float *log_scale_in, float *pose_out,
                                     float *log_scale_out) {
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_window_queued: need 1 <= 2*B*S <= max_pairs");
    int rc = check_common(h, o, 2 * B * S);
    if (rc) return rc;
    if (!tgt || !srcs || !depth_t || !depth_s || !K || !pose_in || !pose_out) return fail(h, TCSFM_E_ARG, "tcsfm_refine_window_queued: NULL argument");
    if (o->host_ptrs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_window_queued: device pointers only");
    if (o->refine != TCSFM_REFINE_POSE_SCALE) { log_scale_in = nullptr; log_scale_out = nullptr; }
    // the REFERENCE rule couples the windows of a call through its batch normalisers: such calls are never merged with others
    // (a profile session does not stop the merging: the merged launches are bracketed like any other -- bench.py's roofline.timed_mode)
    const bool mergeable = h->coal_max > 1 && o->window_rule == TCSFM_WINDOW_PAIR && !h->trace_bits && !h->trace_decide;
    DeviceGuard dev_guard(h->device);
    if ((rc = check_intrinsics(h, o, K, B))) return rc;            // (blocking only the first time a pointer is seen)
    if (!h->pending.empty() && (h->pend_dense || memcmp(&h->pend_opts, o, sizeof(*o)) != 0 || h->pend_B != B || h->pend_S != S ||
                                (long long)2 * B * S * ((long long)h->pending.size() + 1) > h->max_pairs))
        if ((rc = flush_pending(h))) return rc;
    h->pend_opts = *o; h->pend_B = B; h->pend_S = S; h->pend_dense = 0;
    h->pending.push_back({tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, nullptr, log_scale_in, log_scale_out});
    if (!mergeable || (int)h->pending.size() >= h->coal_max || (long long)2 * B * S * ((long long)h->pending.size() + 1) > h->max_pairs)
        return flush_pending(h);
    return TCSFM_OK;
}

// the dense counterpart of tcsfm_refine_window_queued: per-pair Gauss-Newton dense calls with ONE source per target are merged (k_pack_coal,
// per-call pose and depth outputs through the pointer table); every other dense call flushes what is waiting and runs at once
int tcsfm_refine_dense_window_queued(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs, const float *depth_t,
                                     const float *depth_s, const float *K, const float *pose_in, float *pose_out, float *depth_out) {
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window_queued: need 1 <= 2*B*S <= max_pairs");
    int rc = check_common(h, o, 2 * B * S);
    if (rc) return rc;
    if (!tgt || !srcs || !depth_t || !depth_s || !K || !pose_in || !pose_out || !depth_out) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window_queued: NULL argument");
    if (o->host_ptrs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window_queued: device pointers only");
    const bool plain = h->coal_max > 1 && o->solver == TCSFM_SOLVER_GN && o->n_iters >= 1 && o->param == TCSFM_PARAM_SE3 && !h->trace_bits && !h->trace_decide;
    // (round 5) the reference-loss mode with fixed source maps merges too: every call is a normaliser group of its own inside the merged sequence
    const bool merge_ref = plain && o->window_rule == TCSFM_WINDOW_REFERENCE && S <= JMAXS && o->free_source_depths == 0 && o->prior_init >= 0.f && o->w_smooth >= 0.f &&
                           o->w_pose_consist == 0.f && (o->depth_param == TCSFM_DEPTH_FULL || (o->depth_param == TCSFM_DEPTH_QUARTER && h->H % 4 == 0 && h->W % 4 == 0)) &&
                           o->min_depth > 0 && o->max_depth > o->min_depth;
    const bool mergeable = merge_ref || (plain && S == 1 && o->window_rule == TCSFM_WINDOW_PAIR && o->w_dc == 0.f);
    DeviceGuard dev_guard(h->device);
    if (!mergeable) {
        if ((rc = flush_pending(h))) return rc;
        return tcsfm_refine_dense_window(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, depth_out, nullptr);
    }
    if ((rc = check_intrinsics(h, o, K, B))) return rc;            // (blocking only the first time a pointer is seen)
    if (!h->pending.empty() && (!h->pend_dense || memcmp(&h->pend_opts, o, sizeof(*o)) != 0 || h->pend_B != B || h->pend_S != S ||
                                (long long)2 * B * S * ((long long)h->pending.size() + 1) > h->max_pairs))
        if ((rc = flush_pending(h))) return rc;
    h->pend_opts = *o; h->pend_B = B; h->pend_S = S; h->pend_dense = 1;
    h->pending.push_back({tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, depth_out, nullptr, nullptr});
    if ((int)h->pending.size() >= h->coal_max || (long long)2 * B * S * ((long long)h->pending.size() + 1) > h->max_pairs)
        return flush_pending(h);
    return TCSFM_OK;
}

static int linearize_dense_window_impl(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                       const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *depth0,
                                       double *scal_out, double *g_pose_out, float *g_rho_out, float *g_rho_src_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (B < 1 || S < 1 || S > JMAXS || (long long)2 * B * S > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_linearize_dense_window: need 1 <= S <= 3 and 2*B*S <= max_pairs");
    const int N = 2 * B * S;
    int rc = check_common(h, o, N);
    if (rc) return rc;
    if (!tgt || !srcs || !depth_t || !depth_s || !K || !pose || !scal_out || !g_pose_out || !g_rho_out) return fail(h, TCSFM_E_ARG, "tcsfm_linearize_dense_window: NULL argument");
    if (!(o->min_depth > 0 && o->max_depth > o->min_depth) || !(o->prior_init >= 0.f)) return fail(h, TCSFM_E_ARG, "min_depth / max_depth / prior_init invalid");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if ((rc = check_intrinsics(h, o, K, B))) return rc;
    const size_t hw = (size_t)h->H * h->W;
    const float *d_tgt, *d_src, *d_dt, *d_ds, *d_K, *d_pose;
    if ((rc = to_dev(h, o, 0, tgt, (size_t)B * 3 * hw, &d_tgt))) return rc;
    if ((rc = to_dev(h, o, 1, srcs, (size_t)S * B * 3 * hw, &d_src))) return rc;
    if ((rc = to_dev(h, o, 2, depth_t, (size_t)B * hw, &d_dt))) return rc;
    if ((rc = to_dev(h, o, 3, depth_s, (size_t)S * B * hw, &d_ds))) return rc;
    if ((rc = to_dev(h, o, 4, K, (size_t)B * 9, &d_K))) return rc;
    if ((rc = to_dev(h, o, 5, pose, (size_t)N * 6, &d_pose))) return rc;
    const float *d_d0 = nullptr;
    if (depth0 && (rc = to_dev(h, o, 6, depth0, (size_t)B * hw, &d_d0))) return rc;
    float *d_g, *d_gs = nullptr;
    if ((rc = out_dev(h, o, 7, g_rho_out, (size_t)B * hw, &d_g))) return rc;
    if (g_rho_src_out && (rc = out_dev(h, o, 8, g_rho_src_out, (size_t)S * B * hw, &d_gs))) return rc;
    tcsfm_opts oo = *o;
    oo.n_iters = 1; oo.solver = TCSFM_SOLVER_GN;
    DrefExport ex{scal_out, g_pose_out, d_g, d_d0, d_gs};
    rc = S == 1 ? dense_ref_run<1>(h, &oo, B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose, nullptr, nullptr, nullptr, nullptr, &ex)
       : S == 2 ? dense_ref_run<2>(h, &oo, B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose, nullptr, nullptr, nullptr, nullptr, &ex)
                : dense_ref_run<3>(h, &oo, B, d_tgt, d_src, d_dt, d_ds, d_K, d_pose, nullptr, nullptr, nullptr, nullptr, &ex);
    if (rc) return rc;
    if ((rc = copy_back(h, o, g_rho_out, d_g, (size_t)B * hw))) return rc;
    if (g_rho_src_out && (rc = copy_back(h, o, g_rho_src_out, d_gs, (size_t)S * B * hw))) return rc;
    if (o->host_ptrs) HIPCHK(h, hipStreamSynchronize(h->stream));
    return TCSFM_OK;
}

int tcsfm_linearize_dense_window(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                 const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *depth0,
                                 double *scal_out, double *g_pose_out, float *g_rho_out) {
    return linearize_dense_window_impl(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose, depth0, scal_out, g_pose_out, g_rho_out, nullptr);
}
int tcsfm_linearize_dense_window_sources(tcsfm_handle h, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                         const float *depth_t, const float *depth_s, const float *K, const float *pose, const float *depth0,
                                         double *scal_out, double *g_pose_out, float *g_rho_out, float *g_rho_src_out) {
    if (!g_rho_src_out) return h ? fail(h, TCSFM_E_ARG, "tcsfm_linearize_dense_window_sources: NULL argument") : TCSFM_E_ARG;
    return linearize_dense_window_impl(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose, depth0, scal_out, g_pose_out, g_rho_out, g_rho_src_out);
}

// ---- lane self-probe.  MEASURED HAZARD (round 4): how well the streams of a process run side by side depends on the order
// in which the process created them -- a handle created before the process's first device work gave four lanes that were SLOWER than one
// (10 k against 13.5 k frame-pairs/s; kernels of overlapping lanes 40-55 us instead of 11).  The mechanism inside the runtime was not
// identified, so the library measures the real thing: B = 1 refinements (the handle's own kernels, 4 Gauss-Newton iterations) on stand-in
// images, first one after the other on the handle's stream, then round-robin over all lanes -- a stand-in kernel on two streams does NOT
// show the state (it overlapped fine in a process whose four lanes ran at 0.72 of one: the first version of this probe).  Lanes that do
// not beat one stream are switched off: their calls run on the handle's own stream.  ~5 ms, once per tcsfm_set_lanes.
static int probe_lanes(tcsfm_ctx *h) {
    h->lanes_serial = false;
    h->lane_probe[0] = h->lane_probe[1] = 0.f;
    if (h->lanes.empty()) return TCSFM_OK;
    if (const char *e = getenv("TCSFM_LANE_PROBE")) { if (atoi(e) == 0) return TCSFM_OK; }
    const size_t hw = (size_t)h->H * h->W;
    const int L = (int)h->lanes.size() + 1;
    float *buf = nullptr;                                       // tgt | src [3 H W each], depth_t | depth_s [H W each], K [9], pose in [12], pose out [L][12]
    const size_t nfl = 8 * hw + 9 + 12 + (size_t)L * 12;
    HIPCHK(h, hipMalloc((void **)&buf, nfl * sizeof(float)));
    float *tgt = buf, *src = buf + 3 * hw, *dt = buf + 6 * hw, *ds = buf + 7 * hw, *K = buf + 8 * hw, *pin = K + 9, *pout = pin + 12;
    {
        std::vector<float> img(8 * hw);
        for (size_t i = 0; i < 6 * hw; i++) { const size_t p = i % hw; img[i] = 0.5f + 0.25f * sinf(0.05f * (float)(p % h->W) + 0.3f * (float)(i / hw)) * cosf(0.07f * (float)(p / h->W)); }
        for (size_t i = 6 * hw; i < 8 * hw; i++) img[i] = 0.5f + 0.001f * (float)((i % hw) / h->W);
        const float Kh[9] = {0.58f * h->W, 0.f, 0.5f * h->W, 0.f, 1.9f * h->H, 0.5f * h->H, 0.f, 0.f, 1.f};
        const float ph[12] = {0.002f, -0.001f, -0.03f, 0.001f, -0.002f, 0.0005f, -0.002f, 0.001f, 0.03f, -0.001f, 0.002f, -0.0005f};
        HIPCHK(h, hipMemcpy(buf, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(K, Kh, sizeof(Kh), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(pin, ph, sizeof(ph), hipMemcpyHostToDevice));
    }
    tcsfm_opts o; tcsfm_default_opts(&o);
    hipStream_t keep = h->stream;
    h->stream = h->own_stream;
    int rc = TCSFM_OK;
    auto run = [&](int nl, int calls, float *ms) -> int {
        for (int pass = 0; pass < 2 && !rc; pass++) {           // (first pass: warm-up -- scratch allocations, intrinsics check)
            HIPCHK(h, hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < calls && !rc; k++)
                rc = tcsfm_refine_window_async(h, k % nl, &o, 1, 1, tgt, src, dt, ds, K, pin, nullptr, pout + (size_t)(k % nl) * 12, nullptr, nullptr);
            HIPCHK(h, hipDeviceSynchronize());
            *ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return rc;
    };
    const int calls = 24;
    if (h->max_pairs >= 2) {
        rc = run(1, calls, &h->lane_probe[0]);
        if (!rc) rc = run(L, calls, &h->lane_probe[1]);
    }
    h->stream = keep;
    (void)hipFree(buf);
    k_forget(h);
    for (tcsfm_ctx *c : h->lanes) k_forget(c);
    if (rc) return rc;
    if (h->lane_probe[1] > 0.95f * h->lane_probe[0] && h->lane_probe[0] > 0.f) {
        h->lanes_serial = true;
        fprintf(stderr, "tcsfm: lanes do not overlap in this process (probe: %d B=1 refinements take %.0f us on one stream, %.0f us over %d lanes): lane "
                        "calls run on the handle's own stream; use the queued calls (tcsfm_set_coalesce) to keep the chip busy\n", calls,
                h->lane_probe[0] * 1e3, h->lane_probe[1] * 1e3, L);
    }
    return TCSFM_OK;
}

int tcsfm_debug_check_guards(int *n_allocations, int *n_damaged) {
    if (n_allocations) *n_allocations = -1;
    if (n_damaged) *n_damaged = 0;
    if (!tcguard::on()) return TCSFM_OK;
    if (hipDeviceSynchronize() != hipSuccess) return TCSFM_E_HIP;
    std::lock_guard<std::mutex> lk(tcguard::mtx());
    int bad_allocs = 0;
    for (auto &r : tcguard::recs()) {
        long off = 0;
        const long bad = tcguard::damaged(r, &off);
        if (bad < 0) return TCSFM_E_HIP;
        if (bad > 0) {
            bad_allocs++;
            if (!r.reported)
                fprintf(stderr, "tcsfm guard: allocation of %zu bytes (tcsfm_api.hip:%d): %ld band bytes overwritten, first at offset %ld of the allocation\n", r.bytes, r.line, bad, off);
            r.reported = true;
        }
    }
    if (n_allocations) *n_allocations = (int)tcguard::recs().size();
    if (n_damaged) *n_damaged = bad_allocs;
    return TCSFM_OK;
}

int tcsfm_debug_guard_selftest(int *detected) {
    if (detected) *detected = -1;
    if (!tcguard::on()) return TCSFM_OK;
    char *p = nullptr;
    if (hipMalloc(&p, 1000) != hipSuccess) return TCSFM_E_HIP;
    if (hipMemset(p + 1000, 0, 4) != hipSuccess) return TCSFM_E_HIP;      // four bytes past the end
    int found = 0;
    {
        std::lock_guard<std::mutex> lk(tcguard::mtx());
        for (auto &r : tcguard::recs())
            if (r.base + tcguard::kBand == p) { long off = 0; found = tcguard::damaged(r, &off) == 4 && off == 1000; r.reported = true; }
    }
    (void)hipFree(p);
    if (detected) *detected = found;
    return TCSFM_OK;
}

int tcsfm_lane_probe(tcsfm_handle h, int *serial, float *one_stream_ms, float *two_streams_ms) {
    if (!h) return TCSFM_E_ARG;
    if (serial) *serial = h->lanes_serial ? 1 : 0;
    if (one_stream_ms) *one_stream_ms = h->lane_probe[0];
    if (two_streams_ms) *two_streams_ms = h->lane_probe[1];
    return TCSFM_OK;
}

int tcsfm_set_lanes(tcsfm_handle h, int n_lanes) {
    if (!h) return TCSFM_E_ARG;
    if (n_lanes < 1 || n_lanes > 8) return fail(h, TCSFM_E_ARG, "tcsfm_set_lanes: 1 <= n_lanes <= 8");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = flush_pending(h)) return rc_;
    if (int rc_ = join_coalesce_lanes(h)) return rc_;
    if ((int)h->lanes.size() + 1 > n_lanes) HIPCHK(h, hipStreamSynchronize(h->stream));     // (a lane about to go may still run a merged sequence)
    while ((int)h->lanes.size() + 1 > n_lanes) { tcsfm_destroy(h->lanes.back()); h->lanes.pop_back(); }
    h->coal_lanes = std::min(h->coal_lanes, n_lanes);
    while ((int)h->lanes.size() + 1 < n_lanes) {
        tcsfm_ctx *c = nullptr;
        int rc = tcsfm_create(&c, h->device, h->H, h->W, h->max_pairs);
        if (rc) return fail(h, rc, "tcsfm_set_lanes: could not create a lane");
        hipError_t e = hipEventCreateWithFlags(&c->in_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming);
        if (e != hipSuccess) { tcsfm_destroy(c); return fail(h, TCSFM_E_HIP, "tcsfm_set_lanes: hipEventCreate failed"); }
        c->graph_slots = h->graph_slots;
        h->lanes.push_back(c);
    }
    return probe_lanes(h);
}

int tcsfm_set_graph_replay(tcsfm_handle h, int max_graphs) {
    if (!h) return TCSFM_E_ARG;
    if (max_graphs < 0 || max_graphs > 64) return fail(h, TCSFM_E_ARG, "tcsfm_set_graph_replay: 0 <= max_graphs <= 64");
    DeviceGuard dev_guard(h->device);
    h->graph_slots = max_graphs;
    if ((int)h->graphs.size() > max_graphs) drop_graphs(h);
    for (tcsfm_ctx *c : h->lanes) { c->graph_slots = max_graphs; if ((int)c->graphs.size() > max_graphs) drop_graphs(c); }
    return TCSFM_OK;
}

int tcsfm_graph_replay_counts(tcsfm_handle h, int *captures, int *replays) {
    if (!h) return TCSFM_E_ARG;
    int c_ = h->graph_captures, r_ = h->graph_replays;
    for (tcsfm_ctx *c : h->lanes) { c_ += c->graph_captures; r_ += c->graph_replays; }
    if (captures) *captures = c_;
    if (replays) *replays = r_;
    return TCSFM_OK;
}

static tcsfm_ctx *lane_of(tcsfm_ctx *h, int lane) {
    if (!h || lane < 0 || lane > (int)h->lanes.size()) return nullptr;
    return lane == 0 ? h : h->lanes[lane - 1];
}

int tcsfm_refine_window_async(tcsfm_handle h, int lane, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                              const float *depth_t, const float *depth_s, const float *K, const float *pose_in,
                              const float *log_scale_in, float *pose_out, float *log_scale_out, float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    tcsfm_ctx *c = lane_of(h, lane);
    if (!c) return h ? fail(h, TCSFM_E_ARG, "tcsfm_refine_window_async: no such lane (tcsfm_set_lanes)") : TCSFM_E_ARG;
    if (o && o->host_ptrs == 1) return fail(h, TCSFM_E_ARG, "tcsfm_refine_window_async: host_ptrs must be 0 (device) or 2 (pinned host, asynchronous)");
    if (c == h || h->lanes_serial) return tcsfm_refine_window(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out);
    DeviceGuard dev_guard(h->device);
    // the lane starts behind everything queued on the parent's stream so far (the producers of the caller's device buffers) ...
    HIPCHK(h, hipEventRecord(c->in_ev, h->stream));
    HIPCHK(h, hipStreamWaitEvent(c->own_stream, c->in_ev, 0));
    c->stream = c->own_stream;
    int rc = tcsfm_refine_window(c, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, log_scale_in, pose_out, log_scale_out, stats_out);
    if (rc) { h->err = c->err; return rc; }
    // ... and marks its end for tcsfm_lane_wait / tcsfm_lane_synchronize
    HIPCHK(h, hipEventRecord(c->done_ev, c->own_stream));
    return TCSFM_OK;
}

int tcsfm_refine_dense_window_async(tcsfm_handle h, int lane, const tcsfm_opts *o, int B, int S, const float *tgt, const float *srcs,
                                    const float *depth_t, const float *depth_s, const float *K, const float *pose_in, float *pose_out,
                                    float *depth_out, float *stats_out) {
    if (int rc_q = drain_queued(h)) return rc_q;
    tcsfm_ctx *c = lane_of(h, lane);
    if (!c) return h ? fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window_async: no such lane (tcsfm_set_lanes)") : TCSFM_E_ARG;
    if (o && o->host_ptrs == 1) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_window_async: host_ptrs must be 0 (device) or 2 (pinned host, asynchronous)");
    if (c == h || h->lanes_serial) return tcsfm_refine_dense_window(h, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, depth_out, stats_out);
    DeviceGuard dev_guard(h->device);
    HIPCHK(h, hipEventRecord(c->in_ev, h->stream));
    HIPCHK(h, hipStreamWaitEvent(c->own_stream, c->in_ev, 0));
    c->stream = c->own_stream;
    int rc = tcsfm_refine_dense_window(c, o, B, S, tgt, srcs, depth_t, depth_s, K, pose_in, pose_out, depth_out, stats_out);
    if (rc) { h->err = c->err; return rc; }
    HIPCHK(h, hipEventRecord(c->done_ev, c->own_stream));
    return TCSFM_OK;
}

// (the PoseNet section further down)
struct tcsfm_posenet;
static int pose_loop(tcsfm_ctx *h, tcsfm_posenet *pn, int num_iter, int B, int S, const float *tgt, const float *srcs, const float *depth_t,
                     const float *depth_s, const float *K, float *poses_out, float *stacked_out, const WinOff *wo);
static tcsfm_posenet *pn_for_lane(tcsfm_posenet *pn, tcsfm_ctx *c);
static bool pn_usable(const tcsfm_posenet *pn, const tcsfm_ctx *h, int images);
static int pn_max_images(const tcsfm_posenet *pn);

// The reference's sequential driver (run_sequential_optimization.py:186-247) as ONE call: see include/tcsfm.h.  With `pn` the initial
// poses of every window come from the coupled PoseNet loop (train_mono.py:64-80) on the window's lane instead of from the caller.
static int sequence_impl(tcsfm_handle h, const tcsfm_opts *o_in, int T, int S, const float *frames, const float *depths, const float *K,
                         const float *pose_init, tcsfm_posenet *pn, int num_iter, float *pose_init_out, float *pose_out,
                         float *log_scale_out, int ring, int windows_per_call, int target_pos, float *dense_depth_out = nullptr) {
    if (int rc_q = drain_queued(h)) return rc_q;
    if (!h) return TCSFM_E_ARG;
    if (!o_in) return fail(h, TCSFM_E_ARG, "opts is NULL");
    const int N = 2 * S, L = h->lanes_serial ? 1 : (int)h->lanes.size() + 1;
    if (S < 1 || T <= S || N > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: need S >= 1, T > S and 2*S <= max_pairs");
    if (!frames || !depths || !K || (!pose_init && !pn) || !pose_out) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: NULL input");
    if (windows_per_call < 0) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: windows_per_call < 0");
    if (target_pos < -1 || target_pos > S || S > TC_MAX_SRC_OFF) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: target_pos must be -1 or 0..S, S at most 8");
    // position of the target inside a window's S + 1 consecutive frames: the reference's loaders take the middle one
    // (data/kitti_loader.py:271-273: target_idx = int(len / 2), the sources are the others in order)
    const int tp = target_pos < 0 ? (S + 1) / 2 : target_pos;
    WinOff wo;
    wo.on = 1;
    for (int s_ = 0; s_ < TC_MAX_SRC_OFF; s_++) wo.off[s_] = s_ < tp ? s_ : s_ + 1;     // source s = frame w + off[s], relative to the window's first frame
    if (pn && (num_iter < 1 || !pn_usable(pn, h, N) || o_in->depth_is_disp))
        return fail(h, TCSFM_E_ARG, "tcsfm_odometry_sequence: needs a loaded PoseNet of this handle with max_images >= 2*S, num_iter >= 1 and depths (not disparities)");
    tcsfm_opts o = *o_in;
    o.host_ptrs = 0;                                   // the lanes work on the device ring; this call does the staging itself
    const int nwin = T - S;
    // Windows per call: consecutive windows are independent, and the targets (frames w + tp ..) and every source (frames w + off[s] ..)
    // of WB consecutive windows are runs of the ring -- the window form with explicit source positions (WinOff) -- so a lane refines
    // WB windows per call: the kernels fill the chip (26 200 windows/s per call at WB = 8 against 13 900 at WB = 1) and the PoseNet
    // runs on 2 S WB images at a third of the time per image.
    int WB = windows_per_call > 0 ? windows_per_call : 8;
    WB = std::min(WB, std::min(h->max_pairs / N, nwin));
    if (pn) WB = std::min(WB, pn_max_images(pn) / N);
    while (ring > 0 && WB > 1 && ring < WB + S + (ring >= WB + S + 8 ? 4 : 1)) WB--;      // an explicit small ring bounds the windows per call
    int rc = check_common(h, &o, N * WB);
    if (rc) return rc;
    if (K[1] != 0.f || K[3] != 0.f || K[6] != 0.f || K[7] != 0.f || K[8] != 1.f || !(K[0] != 0.f) || !(K[4] != 0.f))
        return fail(h, TCSFM_E_INTRINSICS, "intrinsics must be pinhole [fx 0 cx; 0 fy cy; 0 0 1]");
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    // frames go up in chunks of C (one copy for the images, one for the depths: PCIe runs at 55 GB/s on 8-frame copies, at 36 GB/s
    // on single frames, and the host issues a quarter of the calls); the ring holds a whole number of chunks
    static const int chunk_env = getenv("TCSFM_SEQ_CHUNK") ? std::max(1, atoi(getenv("TCSFM_SEQ_CHUNK"))) : 0;      // (measurement hook)
    // default ring: 4 frames per copy; 8 when a call takes 16 or more windows (the copies then run at 55 instead of 45 GB/s and a call
    // still waits for no more than two of them: 21 800 -> 22 800 windows/s at 16 windows per call, nothing to gain at 8)
    const int C = ring > 0 ? (ring >= WB + S + 8 ? 4 : 1) : (chunk_env ? chunk_env : (WB >= 16 ? 8 : 4));
    const int R = ring > 0 ? (ring / C) * C : ((32 + (L + 1) * WB + S + C - 1) / C) * C;     // frames resident at once
    const int M = WB + S - 1;                              // mirror slots behind the ring: the WB + S frames of a call never wrap
    if (R < S + 2 || R < WB + S + C) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: ring must hold at least windows_per_call + S + 4 frames (S + 2 for one window per call)");
    const size_t hw = (size_t)h->H * h->W;
    // ---- scratch: ring + mirror slots, events, pose staging, WB copies of K
    if (h->seq_slots < R + M) {
        if (h->seq_img) HIPCHK(h, hipFree(h->seq_img));
        if (h->seq_depth) HIPCHK(h, hipFree(h->seq_depth));
        h->seq_img = h->seq_depth = nullptr; h->seq_slots = 0;
        HIPCHK(h, hipMalloc(&h->seq_img, (size_t)(R + M) * 3 * hw * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->seq_depth, (size_t)(R + M) * hw * sizeof(float)));
        if (h->seq_fpack) { HIPCHK(h, hipFree(h->seq_fpack)); h->seq_fpack = nullptr; }       // (re-allocated below for the new slot count)
        if (h->seq_fdepth) { HIPCHK(h, hipFree(h->seq_fdepth)); h->seq_fdepth = nullptr; }
        h->seq_slots = R + M;
    }
    const bool use_cache = dense_depth_out == nullptr;      // (the dense modes rewrite per-pair depth planes: they keep the per-pair pack)
    if (use_cache && !h->seq_fpack) {
        HIPCHK(h, hipMalloc((void **)&h->seq_fpack, (size_t)h->seq_slots * (h->H + 2) * (h->W + 2) * sizeof(float4)));
        HIPCHK(h, hipMalloc((void **)&h->seq_fdepth, (size_t)h->seq_slots * hw * sizeof(float)));
    }
    if (h->seq_pose_cap < (size_t)nwin * N) {
        for (float **q : {&h->seq_pose_in, &h->seq_pose_out, &h->seq_ls_out})
            if (*q) { HIPCHK(h, hipFree(*q)); *q = nullptr; }
        h->seq_pose_cap = 0;
        HIPCHK(h, hipMalloc(&h->seq_pose_in, (size_t)nwin * N * 6 * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->seq_pose_out, (size_t)nwin * N * 6 * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->seq_ls_out, (size_t)nwin * N * sizeof(float)));
        h->seq_pose_cap = (size_t)nwin * N;
    }
    const bool dense = dense_depth_out != nullptr;
    if (dense && h->seq_dense_cap < (size_t)nwin * N * hw) {
        if (h->seq_dense) HIPCHK(h, hipFree(h->seq_dense));
        h->seq_dense = nullptr; h->seq_dense_cap = 0;
        HIPCHK(h, hipMalloc(&h->seq_dense, (size_t)nwin * N * hw * sizeof(float)));
        h->seq_dense_cap = (size_t)nwin * N * hw;
    }
    if (dense && h->seq_dense_tmp_cap < (size_t)WB * N * hw) {
        if (h->seq_dense_tmp) HIPCHK(h, hipFree(h->seq_dense_tmp));
        h->seq_dense_tmp = nullptr; h->seq_dense_tmp_cap = 0;
        HIPCHK(h, hipMalloc(&h->seq_dense_tmp, (size_t)WB * N * hw * sizeof(float)));
        h->seq_dense_tmp_cap = (size_t)WB * N * hw;
    }
    if (h->seq_K_n < WB) {
        if (h->seq_K) HIPCHK(h, hipFree(h->seq_K));
        h->seq_K = nullptr; h->seq_K_n = 0;
        HIPCHK(h, hipMalloc(&h->seq_K, (size_t)WB * 9 * sizeof(float)));
        h->seq_K_n = WB;
    }
    if (!h->seq_copy) {   // high priority: a hardware queue outside the pool the normal-priority streams share (a copy stream that lands in
        int lo = 0, hi = 0;   // a lane's queue parks its slot-recycling waits in front of that lane's kernels), and copies go first anyway
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(h, hipStreamCreateWithPriority(&h->seq_copy, hipStreamNonBlocking, hi));
    }
    const size_t n_done = (size_t)2 * R + 2;               // a call's event is re-recorded long after its slots were recycled
    const size_t ND = n_done - 1;
    while (h->seq_copied.size() < (size_t)R) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->seq_copied.push_back(e); }
    while (use_cache && h->seq_raw.size() < (size_t)R) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->seq_raw.push_back(e); }
    if (use_cache && !h->seq_pack) HIPCHK(h, hipStreamCreateWithFlags(&h->seq_pack, hipStreamNonBlocking));
    while (h->seq_done.size() < n_done) { hipEvent_t e; HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->seq_done.push_back(e); }
    // Poses live on the device per CALL in the stacked order of the window form (forward pairs (s, b), then inverse pairs); the
    // caller's arrays are per window.  at(): offset (in pairs) of pair j of window w inside the staging arrays.
    auto at = [&](int w, int j) -> size_t {
        const int c0 = (w / WB) * WB, nbw = std::min(WB, nwin - c0), b = w - c0;
        return (size_t)c0 * N + (j < S ? (size_t)j * nbw + b : (size_t)S * nbw + (size_t)(j - S) * nbw + b);
    };
    // ---- small inputs: one copy each, on the copy stream; every lane waits for them once
    hipStream_t cs = h->seq_copy;
    std::vector<float> Kh((size_t)WB * 9), stage;
    for (int b = 0; b < WB; b++) memcpy(&Kh[(size_t)b * 9], K, 9 * sizeof(float));
    HIPCHK(h, hipMemcpyAsync(h->seq_K, Kh.data(), Kh.size() * sizeof(float), hipMemcpyHostToDevice, cs));
    if (!pn) {
        const float *src = pose_init;
        if (WB > 1) {
            stage.resize((size_t)nwin * N * 6);
            for (int w = 0; w < nwin; w++)
                for (int j = 0; j < N; j++) memcpy(&stage[at(w, j) * 6], pose_init + ((size_t)w * N + j) * 6, 6 * sizeof(float));
            src = stage.data();
        }
        HIPCHK(h, hipMemcpyAsync(h->seq_pose_in, src, (size_t)nwin * N * 6 * sizeof(float), hipMemcpyHostToDevice, cs));
    }
    hipEvent_t small_ev = h->seq_done[n_done - 1];
    HIPCHK(h, hipEventRecord(small_ev, cs));
    HIPCHK(h, hipEventSynchronize(small_ev));              // Kh / stage are host temporaries (pageable): gone from here on
    std::vector<tcsfm_ctx *> lane(L);
    std::vector<tcsfm_posenet *> net(L, nullptr);
    std::vector<hipStream_t> ls(L);
    for (int l = 0; l < L; l++) {
        lane[l] = l == 0 ? h : h->lanes[l - 1];
        if (pn && !(net[l] = pn_for_lane(pn, lane[l]))) return fail(h, TCSFM_E_NOMEM, "tcsfm_odometry_sequence: no memory for a lane's PoseNet activations");
        ls[l] = l == 0 ? h->stream : lane[l]->own_stream;
        lane[l]->stream = ls[l];
        k_add(lane[l], h->seq_K, WB);        // validated on the host above
    }
    std::vector<long long> slot_reader(R, -1);             // last CALL that reads the frame in this slot (-1: none pending)
    const int np = np_of(&o);
    const int ahead = S + L * WB + C;                      // frames kept in flight ahead of the first window of the call being issued
    const long long depth = std::max(2 * L, 16 / WB);      // calls the host may run ahead of the GPU
    int nxt = 0;                                           // first frame of the next copy chunk
    long long ci = 0;                                      // call index
    for (int c0 = 0; c0 < nwin; c0 += WB, ci++) {
        const int nbw = std::min(WB, nwin - c0);
        // the host stays a bounded number of calls ahead of the GPU: a deeper backlog buys nothing, and with one the runtime was seen to
        // block a single hipMemcpyAsync of the copy stream for 7 ms while the lanes ran dry behind it
        if (ci >= depth) HIPCHK(h, hipEventSynchronize(h->seq_done[(ci - depth) % ND]));
        // a chunk may go up once every window that reads the frames it overwrites has been ISSUED (their events exist): nxt - R + C - 1 < c0
        while (nxt < T && nxt <= c0 + nbw - 1 + ahead && nxt + C - 1 - R < c0) {
            const int slot = nxt % R, nf = T - nxt < C ? T - nxt : C;
            long long last = -1;                           // the slots' previous frames may still be read: the copy waits for their readers --
            for (int k = 0; k < nf; k++) { if (slot_reader[slot + k] > last) last = slot_reader[slot + k]; slot_reader[slot + k] = -1; }
            for (long long r = last; r >= 0 && r > last - L; r--)     // -- the latest of those calls on every lane (a lane's stream is in order)
                HIPCHK(h, hipStreamWaitEvent(cs, h->seq_done[r % ND], 0));
            HIPCHK(h, hipMemcpyAsync(h->seq_img + (size_t)slot * 3 * hw, frames + (size_t)nxt * 3 * hw, (size_t)nf * 3 * hw * sizeof(float), hipMemcpyHostToDevice, cs));
            HIPCHK(h, hipMemcpyAsync(h->seq_depth + (size_t)slot * hw, depths + (size_t)nxt * hw, (size_t)nf * hw * sizeof(float), hipMemcpyHostToDevice, cs));
            // mirror of the first M slots behind the ring (the frames of a call never wrap).  The frames cross PCIe ONCE: raw mirrors are
            // device-to-device copies, and with the pack cache only the PoseNet reads raw frames -- otherwise the mirror exists as packs only
            const int nm = slot < M ? (nf < M - slot ? nf : M - slot) : 0;
            if (nm > 0 && (!use_cache || pn)) {
                HIPCHK(h, hipMemcpyAsync(h->seq_img + (size_t)(R + slot) * 3 * hw, h->seq_img + (size_t)slot * 3 * hw, (size_t)nm * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, cs));
                HIPCHK(h, hipMemcpyAsync(h->seq_depth + (size_t)(R + slot) * hw, h->seq_depth + (size_t)slot * hw, (size_t)nm * hw * sizeof(float), hipMemcpyDeviceToDevice, cs));
            }
            if (use_cache) {      // pack the frames that just landed (and their mirrors), once -- on the PACK stream, so that the next chunk's
                                  // copy does not queue behind the kernel (a sequence with S = 1 is PCIe-bound: the copy stream is its critical path)
                HIPCHK(h, hipEventRecord(h->seq_raw[slot / C], cs));
                HIPCHK(h, hipStreamWaitEvent(h->seq_pack, h->seq_raw[slot / C], 0));
                FramePackParams Fp;
                Fp.H = h->H; Fp.W = h->W; Fp.depth_is_disp = o.depth_is_disp;
                Fp.min_disp = o.depth_is_disp ? 1.f / o.max_depth : 0.f; Fp.max_disp = o.depth_is_disp ? 1.f / o.min_depth : 0.f;
                auto pack = [&](int from, int to, int count) {
                    Fp.img = h->seq_img + (size_t)from * 3 * hw; Fp.depth = h->seq_depth + (size_t)from * hw;
                    Fp.fpack = h->seq_fpack + (size_t)to * (h->H + 2) * (h->W + 2); Fp.fdepth = h->seq_fdepth + (size_t)to * hw;
                    hipLaunchKernelGGL(k_frame_pack, dim3((unsigned)((hw + 255) / 256), count), dim3(256), 0, h->seq_pack, Fp);
                };
                pack(slot, slot, nf);
                if (nm > 0) pack(slot, R + slot, nm);
                HIPCHK(h, hipEventRecord(h->seq_copied[slot / C], h->seq_pack));
            } else {
                HIPCHK(h, hipEventRecord(h->seq_copied[slot / C], cs));
            }
            nxt += nf;
        }
        const int l = (int)(ci % L), s0 = c0 % R;
        for (int k = c0 / C; k <= (c0 + nbw - 1 + S) / C; k++) HIPCHK(h, hipStreamWaitEvent(ls[l], h->seq_copied[(k * C % R) / C], 0));
        tcsfm_ctx *c = lane[l];
        const float *tg = h->seq_img + (size_t)(s0 + tp) * 3 * hw, *sr = h->seq_img + (size_t)s0 * 3 * hw;      // sources: by position (wo)
        const float *dt = h->seq_depth + (size_t)(s0 + tp) * hw, *ds = h->seq_depth + (size_t)s0 * hw;
        float *p_in = h->seq_pose_in + (size_t)c0 * N * 6, *p_out = h->seq_pose_out + (size_t)c0 * N * 6;
        if (pn) {           // initial poses of these windows: PoseNet -> warp -> PoseNet correction, num_iter times, on the lane
            rc = pose_loop(c, net[l], num_iter, nbw, S, tg, sr, dt, ds, h->seq_K, p_in, nullptr, &wo);
            if (rc) { if (c != h) h->err = c->err; break; }
        }
        if (dense) {
            // the call's maps come out stacked [pair index][window]; a small kernel puts them in the caller's per-window order, and ONE
            // copy per call takes them to the host on the lane's stream (asynchronous when the destination is pinned)
            if (c->seq_dense_cap < (size_t)WB * N * hw) {
                if (c->seq_dense && c != h) HIPCHK(h, hipFree(c->seq_dense));
                if (c != h) { c->seq_dense = nullptr; c->seq_dense_cap = 0; HIPCHK(h, hipMalloc(&c->seq_dense, (size_t)WB * N * hw * sizeof(float))); c->seq_dense_cap = (size_t)WB * N * hw; }
            }
            float *tmp = c == h ? h->seq_dense_tmp : c->seq_dense;
            rc = dense_impl(c, &o, N * nbw, nbw, S, tg, sr, dt, ds, h->seq_K, p_in, p_out, tmp, nullptr, &wo);
            if (!rc) {
                float *ordered = h->seq_dense + (size_t)c0 * N * hw;
                hipLaunchKernelGGL(k_maps_to_window_order, dim3((unsigned)((hw + 255) / 256), N * nbw), dim3(256), 0, ls[l], (const float *)tmp, ordered, nbw, N, (int)hw);
                HIPCHK(h, hipMemcpyAsync(dense_depth_out + (size_t)c0 * N * hw, ordered, (size_t)nbw * N * hw * sizeof(float), hipMemcpyDeviceToHost, ls[l]));
            }
        } else {
            FrameCache fcache = {h->seq_fpack, h->seq_fdepth, s0, tp};
            rc = refine_impl(c, &o, N * nbw, nbw, S, tg, sr, dt, ds, h->seq_K, p_in, nullptr, p_out,
                             np == 7 ? h->seq_ls_out + (size_t)c0 * N : nullptr, nullptr, &wo, use_cache ? &fcache : nullptr);
        }
        if (rc) { if (c != h) h->err = c->err; break; }
        HIPCHK(h, hipEventRecord(h->seq_done[ci % ND], ls[l]));
        for (int k = 0; k < nbw + S; k++) slot_reader[(c0 + k) % R] = ci;
    }
    // ---- drain: every lane, then the results in one copy each
    for (int l = 0; l < L; l++) {
        hipError_t e = hipStreamSynchronize(ls[l]);
        if (e != hipSuccess && !rc) { h->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); rc = TCSFM_E_HIP; }
    }
    (void)hipStreamSynchronize(cs);
    if (h->seq_pack) (void)hipStreamSynchronize(h->seq_pack);
    if (rc) return rc;
    for (int l = 0; l < L; l++)
        if (int rc_ = pending_error(lane[l])) { if (lane[l] != h) h->err = lane[l]->err; return rc_; }
    auto fetch = [&](float *dst, const float *dev, int per_pair) -> int {     // device (per call, stacked) -> caller (per window)
        if (WB == 1) { HIPCHK(h, hipMemcpy(dst, dev, (size_t)nwin * N * per_pair * sizeof(float), hipMemcpyDeviceToHost)); return TCSFM_OK; }
        stage.resize((size_t)nwin * N * per_pair);
        HIPCHK(h, hipMemcpy(stage.data(), dev, stage.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int w = 0; w < nwin; w++)
            for (int j = 0; j < N; j++) memcpy(dst + ((size_t)w * N + j) * per_pair, &stage[at(w, j) * per_pair], per_pair * sizeof(float));
        return TCSFM_OK;
    };
    if ((rc = fetch(pose_out, h->seq_pose_out, 6))) return rc;
    if (pose_init_out && (rc = fetch(pose_init_out, h->seq_pose_in, 6))) return rc;
    if (log_scale_out && np == 7 && !dense && (rc = fetch(log_scale_out, h->seq_ls_out, 1))) return rc;
    return TCSFM_OK;
}

int tcsfm_refine_sequence(tcsfm_handle h, const tcsfm_opts *o, int T, int S, const float *frames, const float *depths, const float *K,
                          const float *pose_init, float *pose_out, float *log_scale_out, int ring, int windows_per_call, int target_pos) {
    if (h && !pose_init) return fail(h, TCSFM_E_ARG, "tcsfm_refine_sequence: NULL input");
    return sequence_impl(h, o, T, S, frames, depths, K, pose_init, nullptr, 0, nullptr, pose_out, log_scale_out, ring, windows_per_call, target_pos);
}

int tcsfm_odometry_sequence(tcsfm_handle h, tcsfm_posenet *pn, int num_iter, const tcsfm_opts *o, int T, int S, const float *frames,
                            const float *depths, const float *K, float *pose_init_out, float *pose_out, float *log_scale_out, int ring,
                            int windows_per_call, int target_pos) {
    if (h && !pn) return fail(h, TCSFM_E_ARG, "tcsfm_odometry_sequence: NULL PoseNet");
    return sequence_impl(h, o, T, S, frames, depths, K, nullptr, pn, num_iter, pose_init_out, pose_out, log_scale_out, ring, windows_per_call, target_pos);
}

int tcsfm_refine_dense_sequence(tcsfm_handle h, const tcsfm_opts *o, int T, int S, const float *frames, const float *depths, const float *K,
                                const float *pose_init, float *pose_out, float *depth_out, int ring, int windows_per_call, int target_pos) {
    if (h && (!pose_init || !depth_out)) return fail(h, TCSFM_E_ARG, "tcsfm_refine_dense_sequence: NULL input");
    return sequence_impl(h, o, T, S, frames, depths, K, pose_init, nullptr, 0, nullptr, pose_out, nullptr, ring, windows_per_call, target_pos, depth_out);
}

int tcsfm_lane_wait(tcsfm_handle h, int lane) {
    tcsfm_ctx *c = lane_of(h, lane);
    if (!c) return h ? fail(h, TCSFM_E_ARG, "tcsfm_lane_wait: no such lane") : TCSFM_E_ARG;
    if (c == h) return TCSFM_OK;
    DeviceGuard dev_guard(h->device);
    HIPCHK(h, hipStreamWaitEvent(h->stream, c->done_ev, 0));     // consumers on the parent's stream see the lane's outputs
    return TCSFM_OK;
}

int tcsfm_lane_event(tcsfm_handle h, int lane, void **event_out) {
    tcsfm_ctx *c = lane_of(h, lane);
    if (!c || !event_out) return h ? fail(h, TCSFM_E_ARG, "tcsfm_lane_event: bad argument") : TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    constexpr size_t kMarks = 64;
    if (c->marks.size() < kMarks) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->marks.push_back(e);
        c->mark_next = c->marks.size() - 1;
    }
    hipEvent_t e = c->marks[c->mark_next];
    c->mark_next = (c->mark_next + 1) % kMarks;
    HIPCHK(h, hipEventRecord(e, (c == h || h->lanes_serial) ? h->stream : c->own_stream));      // (serial lanes: the lane's calls ran on the handle's stream)
    *event_out = (void *)e;
    return TCSFM_OK;
}

int tcsfm_stream_wait_event(tcsfm_handle h, void *hip_stream, void *event) {
    if (!h || !event) return TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    HIPCHK(h, hipStreamWaitEvent((hipStream_t)hip_stream, (hipEvent_t)event, 0));
    return TCSFM_OK;
}

int tcsfm_lane_synchronize(tcsfm_handle h, int lane) {
    tcsfm_ctx *c = lane_of(h, lane);
    if (!c) return h ? fail(h, TCSFM_E_ARG, "tcsfm_lane_synchronize: no such lane") : TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    HIPCHK(h, hipStreamSynchronize((c == h || h->lanes_serial) ? h->stream : c->own_stream));
    if (int rc = pending_error(c)) { h->err = c->err; return rc; }
    if (h->lanes_serial) return pending_error(h);
    return TCSFM_OK;
}

int tcsfm_debug_trace(tcsfm_handle h, uint16_t *bits, int64_t bits_capacity, int32_t *decide, int64_t decide_capacity) {
    if (!h) return TCSFM_E_ARG;
    if ((bits && bits_capacity < 1) || (decide && decide_capacity < 1)) return fail(h, TCSFM_E_ARG, "tcsfm_debug_trace: bad capacity");
    h->trace_bits = bits; h->trace_bits_cap = bits ? bits_capacity : 0;
    h->trace_decide = decide; h->trace_decide_cap = decide ? decide_capacity : 0;
    return TCSFM_OK;
}

// diagnostic: copy the k_solve phase stamps to the host (8 values; zeros unless TCSFM_DEBUG_STAMPS was set at create)
int tcsfm_debug_stamps(tcsfm_handle h, long long out[8]) {
    if (!h || !out) return TCSFM_E_ARG;
    memset(out, 0, 8 * sizeof(long long));
    if (!h->dbg_stamps) return TCSFM_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->dbg_stamps, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return TCSFM_OK;
}

int tcsfm_profile_begin(tcsfm_handle h) {
    if (!h) return TCSFM_E_ARG;
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    if (!h->stamp_buf) {
        h->stamp_cap = (size_t)4 << 20;     // 4 M workgroup stamps = 64 MB: ~8000 launches of the BASELINE config
        HIPCHK(h, hipMalloc((void **)&h->stamp_buf, h->stamp_cap * 2 * sizeof(unsigned long long)));
    }
    h->stamp_used = 0;
    h->stamp_launch.clear();
    for (tcsfm_ctx *c : h->lanes)                       // the lanes are bracketed too: their figures are added in profile_end
        if (int rc = tcsfm_profile_begin(c)) { h->err = c->err; return rc; }
    h->profiling = true;
    h->ev_used = 0;
    h->ev_class.clear();
    return TCSFM_OK;
}

int tcsfm_profile_end(tcsfm_handle h, double ms_sum[3], int64_t launches[3]) {
    if (!h || !ms_sum || !launches) return TCSFM_E_ARG;
    h->profiling = false;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; i++) { ms_sum[i] = 0.0; launches[i] = 0; }
    for (size_t k = 0; k < h->ev_class.size(); k++) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_pool[2 * k], h->ev_pool[2 * k + 1]));
        ms_sum[h->ev_class[k]] += ms;
        launches[h->ev_class[k]]++;
    }
    h->ev_used = 0;
    h->ev_class.clear();
    for (tcsfm_ctx *c : h->lanes) {
        double ms[3]; int64_t n[3];
        if (int rc = tcsfm_profile_end(c, ms, n)) { h->err = c->err; return rc; }
        for (int i = 0; i < 3; i++) { ms_sum[i] += ms[i]; launches[i] += n[i]; }
    }
    return TCSFM_OK;
}

int tcsfm_profile_kernel_time(tcsfm_handle h, double *ms_sum, int64_t *launches) {
    if (!h || !ms_sum || !launches) return TCSFM_E_ARG;
    *ms_sum = 0.0; *launches = 0;
    DeviceGuard dev_guard(h->device);
    if (h->stamp_buf && h->stamp_used > 0) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> st(h->stamp_used * 2);
        HIPCHK(h, hipMemcpy(st.data(), h->stamp_buf, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (const auto &l : h->stamp_launch) {     // duration of a launch = latest workgroup end - earliest workgroup start
            unsigned long long t0 = ~0ull, t1 = 0ull;
            for (size_t w = l.first; w < l.first + l.second; w++) { t0 = st[2 * w] < t0 ? st[2 * w] : t0; t1 = st[2 * w + 1] > t1 ? st[2 * w + 1] : t1; }
            if (t1 > t0) { *ms_sum += (double)(t1 - t0) * 1e-5; (*launches)++; }   // 100 MHz ticks -> ms
        }
    }
    for (tcsfm_ctx *c : h->lanes) {
        double ms = 0.0; int64_t n = 0;
        if (int rc = tcsfm_profile_kernel_time(c, &ms, &n)) { h->err = c->err; return rc; }
        *ms_sum += ms; *launches += n;
    }
    return TCSFM_OK;
}

// [start, end] of every stamped launch of the handle and of its lanes, in 100 MHz ticks of the device's s_memrealtime counter
static int profile_intervals(tcsfm_ctx *h, std::vector<std::pair<unsigned long long, unsigned long long>> &out) {
    DeviceGuard dev_guard(h->device);
    if (h->stamp_buf && h->stamp_used > 0) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> st(h->stamp_used * 2);
        HIPCHK(h, hipMemcpy(st.data(), h->stamp_buf, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (const auto &l : h->stamp_launch) {
            unsigned long long t0 = ~0ull, t1 = 0ull;
            for (size_t w = l.first; w < l.first + l.second; w++) { t0 = st[2 * w] < t0 ? st[2 * w] : t0; t1 = st[2 * w + 1] > t1 ? st[2 * w + 1] : t1; }
            if (t1 > t0) out.push_back({t0, t1});
        }
    }
    for (tcsfm_ctx *c : h->lanes)
        if (int rc = profile_intervals(c, out)) { h->err = c->err; return rc; }
    return TCSFM_OK;
}

int tcsfm_profile_kernel_busy(tcsfm_handle h, double *ms_busy, int64_t *launches) {
    if (!h || !ms_busy || !launches) return TCSFM_E_ARG;
    *ms_busy = 0.0; *launches = 0;
    std::vector<std::pair<unsigned long long, unsigned long long>> iv;
    if (int rc = profile_intervals(h, iv)) return rc;
    std::sort(iv.begin(), iv.end());
    unsigned long long busy = 0, lo = 0, hi = 0;
    bool open = false;
    for (const auto &x : iv) {
        if (open && x.first <= hi) { hi = std::max(hi, x.second); continue; }
        if (open) busy += hi - lo;
        lo = x.first; hi = x.second; open = true;
    }
    if (open) busy += hi - lo;
    *ms_busy = (double)busy * 1e-5;
    *launches = (int64_t)iv.size();
    return TCSFM_OK;
}

// ---- PoseNet (models/pose_models.py:88-147) and the coupled pose loop (train_mono.py:64-80) ---------------------------------
struct tcsfm_posenet {
    tcsfm_ctx *h = nullptr;
    int max_images = 0, loaded = 0;
    PnLayer L[7];
    int nb_cfg[2][7] = {}, ks_cfg[2][7] = {};   // (output-channel blocks per wave, K split) per layer: [0] few images (latency), [1] many
    pn_f4 *w4[7] = {};
    float *bias[7] = {}, *gamma[7] = {}, *beta[7] = {};
    float *act[7] = {}, *scsh[7] = {}, *part[7] = {};
    float *head_w = nullptr, *head_b = nullptr, *raw = nullptr;
    float *in_buf = nullptr;     // [max_images,6,H,W] (tgt * valid | img_rec) written by the warp kernel
    float *pose = nullptr;       // [max_images,6] running pose of the coupled loop
    // tcsfm_odometry_sequence runs the network on the handle's lanes: clone k works on lane k with its own activations and
    // borrows this object's weights
    bool owns_weights = true;
    std::vector<tcsfm_posenet *> clones;
};

void tcsfm_posenet_destroy(tcsfm_posenet *pn) {
    if (!pn) return;
    for (tcsfm_posenet *c : pn->clones) tcsfm_posenet_destroy(c);
    DeviceGuard dev_guard(pn->h->device);
    for (int l = 0; l < 7; l++) {
        void *weights[] = {pn->w4[l], pn->bias[l], pn->gamma[l], pn->beta[l]}, *scratch[] = {pn->act[l], pn->scsh[l], pn->part[l]};
        if (pn->owns_weights) for (void *p : weights) if (p) (void)hipFree(p);
        for (void *p : scratch) if (p) (void)hipFree(p);
    }
    void *weights[] = {pn->head_w, pn->head_b, pn->raw}, *scratch[] = {pn->in_buf, pn->pose};
    if (pn->owns_weights) for (void *p : weights) if (p) (void)hipFree(p);
    for (void *p : scratch) if (p) (void)hipFree(p);
    delete pn;
}

// activations, statistics and loop buffers of one PoseNet instance (layer geometry and work split already filled in)
static hipError_t pn_alloc_scratch(tcsfm_posenet *pn) {
    hipError_t e = hipSuccess;
    const int max_images = pn->max_images;
    for (int l = 0; l < 7 && e == hipSuccess; l++) {
        const PnLayer &L = pn->L[l];
        e = hipMalloc((void **)&pn->act[l], (size_t)L.ksplit * max_images * L.oh * L.ow * L.cout * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&pn->scsh[l], (size_t)max_images * L.cout * 2 * sizeof(float));
        if (e == hipSuccess && (pn->ks_cfg[0][l] == 1 || pn->ks_cfg[1][l] == 1))
            e = hipMalloc((void **)&pn->part[l], (size_t)max_images * std::max((L.oh * L.ow + 63) / 64, L.oh * ((L.ow + 63) / 64)) * L.cout * 2 * sizeof(float));
    }
    if (e == hipSuccess) e = hipMalloc((void **)&pn->in_buf, (size_t)max_images * 6 * pn->h->H * pn->h->W * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&pn->pose, (size_t)max_images * 6 * sizeof(float));
    return e;
}

// the instance that runs `pn`'s network on lane context `c` (lane 0 = pn itself)
static tcsfm_posenet *pn_for_lane(tcsfm_posenet *pn, tcsfm_ctx *c) {
    if (pn->h == c) return pn;
    for (tcsfm_posenet *q : pn->clones)
        if (q->h == c) return q;
    tcsfm_posenet *q = new tcsfm_posenet(*pn);          // geometry, work split, weight pointers
    q->h = c; q->owns_weights = false; q->clones.clear();
    for (int l = 0; l < 7; l++) q->act[l] = q->scsh[l] = q->part[l] = nullptr;
    q->in_buf = q->pose = nullptr;
    if (pn_alloc_scratch(q) != hipSuccess) { tcsfm_posenet_destroy(q); return nullptr; }
    pn->clones.push_back(q);
    return q;
}

static bool pn_usable(const tcsfm_posenet *pn, const tcsfm_ctx *h, int images) { return pn && pn->h == h && pn->loaded && images <= pn->max_images; }
static int pn_max_images(const tcsfm_posenet *pn) { return pn->max_images; }

int tcsfm_posenet_create(tcsfm_handle h, int max_images, tcsfm_posenet **out) {
    if (!h || !out) return TCSFM_E_ARG;
    *out = nullptr;
    if (max_images < 1 || max_images > 4096) return fail(h, TCSFM_E_ARG, "tcsfm_posenet_create: max_images out of range");
    DeviceGuard dev_guard(h->device);
    tcsfm_posenet *pn = new tcsfm_posenet();
    pn->h = h; pn->max_images = max_images;
    static const int chans[8] = {6, 16, 32, 64, 128, 256, 256, 256}, ksz[7] = {7, 5, 3, 3, 3, 3, 3};
    int ih = h->H, iw = h->W;
    hipError_t e = hipSuccess;
    size_t wmax = 0;
    for (int l = 0; l < 7; l++) {
        PnLayer &L = pn->L[l];
        L.cin = chans[l]; L.cout = chans[l + 1]; L.ks = ksz[l]; L.pad = (ksz[l] - 1) / 2;
        L.ih = ih; L.iw = iw; L.oh = (ih + 2 * L.pad - L.ks) / 2 + 1; L.ow = (iw + 2 * L.pad - L.ks) / 2 + 1;
        L.kgroups = l == 0 ? 21 : L.ks * L.ks * L.cin / 16;
        // Work split of a layer = (output-channel blocks of 16 per wave, K split).  A wave's K loop is a serial chain of loads and
        // matrix-core steps, and a window's fwd + inv pair is only 2 images: with 4 channel blocks per wave and K whole the small
        // layers ran on ~100 waves of ~300 dependent MFMAs each (15 us per layer whatever its size).  Two fixed regimes, chosen by
        // the number of images only (results do not depend on anything else):
        //   few images (N <= 4): as few channel blocks per wave as it takes to have ~800 waves for N = 2, then K split until they
        //                        exist or a wave's loop is down to 8 groups;
        //   many images:         up to 4 channel blocks per wave, K split only for the late layers (few output pixels).
        {
            const int pxb = (L.oh * L.ow + 15) / 16, cb = L.cout / 16;
            int nb = std::min(cb, 4), ks = 1;
            while (nb > 1 && pxb * (cb / nb) * 2 < 768) nb /= 2;
            while (ks < 16 && pxb * (cb / nb) * 2 * ks < 768 && L.kgroups / (2 * ks) >= 8) ks *= 2;
            pn->nb_cfg[0][l] = l == 0 ? 1 : nb; pn->ks_cfg[0][l] = l == 0 ? 1 : ks;
            pn->nb_cfg[1][l] = L.cout >= 64 ? 4 : cb;
            pn->ks_cfg[1][l] = L.oh * L.ow <= 512 ? std::min(16, (L.kgroups + 23) / 24) : 1;
        }
        L.ksplit = std::max(pn->ks_cfg[0][l], pn->ks_cfg[1][l]);     // allocation; the launch sets the split it uses
        if (L.oh < 1 || L.ow < 1) { tcsfm_posenet_destroy(pn); return fail(h, TCSFM_E_ARG, "tcsfm_posenet_create: image too small for seven stride-2 layers"); }
        const size_t nw4 = (size_t)L.kgroups * 4 * L.cout;
        if (e == hipSuccess) e = hipMalloc((void **)&pn->w4[l], nw4 * sizeof(pn_f4));
        if (e == hipSuccess) e = hipMalloc((void **)&pn->bias[l], L.cout * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&pn->gamma[l], L.cout * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&pn->beta[l], L.cout * sizeof(float));
        wmax = std::max(wmax, (size_t)L.cout * L.cin * L.ks * L.ks);
        ih = L.oh; iw = L.ow;
    }
    if (e == hipSuccess) e = hipMalloc((void **)&pn->head_w, 6 * 256 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&pn->head_b, 6 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&pn->raw, wmax * sizeof(float));
    if (e == hipSuccess) e = pn_alloc_scratch(pn);
    if (e != hipSuccess) { tcsfm_posenet_destroy(pn); return fail(h, e == hipErrorOutOfMemory ? TCSFM_E_NOMEM : TCSFM_E_HIP, "tcsfm_posenet_create: allocation failed"); }
    *out = pn;
    return TCSFM_OK;
}

int tcsfm_posenet_load(tcsfm_posenet *pn, const float *const conv_w[7], const float *const conv_b[7], const float *const gn_w[7],
                       const float *const gn_b[7], const float *head_w, const float *head_b) {
    if (!pn) return TCSFM_E_ARG;
    tcsfm_ctx *h = pn->h;
    if (!conv_w || !head_w || !head_b) return fail(h, TCSFM_E_ARG, "tcsfm_posenet_load: NULL argument");
    DeviceGuard dev_guard(h->device);
    for (int l = 0; l < 7; l++) {
        const PnLayer &L = pn->L[l];
        if (!conv_w[l]) return fail(h, TCSFM_E_ARG, "tcsfm_posenet_load: NULL convolution weight");
        const size_t nw = (size_t)L.cout * L.cin * L.ks * L.ks;
        HIPCHK(h, hipMemcpyAsync(pn->raw, conv_w[l], nw * sizeof(float), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_pn_prep, dim3(L.cout), dim3(256), 0, h->stream, (const float *)pn->raw, pn->w4[l], L.cin, L.cout, L.ks, l == 0 ? 1 : 0, 1);
        HIPCHK(h, hipStreamSynchronize(h->stream));    // pn->raw is reused by the next layer; loading happens once per model
        std::vector<float> ones(L.cout, 1.f), zeros(L.cout, 0.f);
        HIPCHK(h, hipMemcpy(pn->bias[l], conv_b && conv_b[l] ? conv_b[l] : zeros.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(pn->gamma[l], gn_w && gn_w[l] ? gn_w[l] : ones.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(pn->beta[l], gn_b && gn_b[l] ? gn_b[l] : zeros.data(), L.cout * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemcpy(pn->head_w, head_w, 6 * 256 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(pn->head_b, head_b, 6 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipGetLastError());
    pn->loaded = 1;
    return TCSFM_OK;
}

namespace {
// the seven convolutions + statistics passes + head of one PoseNet evaluation on N samples; the first layer reads
// (imgA | imgB) per sample (strides in floats; window indexing when win_B > 0)
int pn_run(tcsfm_posenet *pn, int N, const float *imgA, long long strideA, const float *imgB, long long strideB, int win_B, int win_S,
           float *pose, int accumulate, float *stacked, int it, int iters, const WinOff *wo = nullptr) {
    tcsfm_ctx *h = pn->h;
    const int cfg = N <= 4 ? 0 : 1;
    for (int l = 0; l < 7; l++) {
        PnLayer L = pn->L[l];
        L.ksplit = pn->ks_cfg[cfg][l];
        const int nb = pn->nb_cfg[cfg][l];
        PnConvParams P;
        memset(&P, 0, sizeof(P));
        P.imgA = imgA; P.imgB = imgB; P.strideA = strideA; P.strideB = strideB; P.win_B = win_B; P.win_S = win_S;
        if (wo) P.win_off = *wo;
        P.in = l > 0 ? pn->act[l - 1] : nullptr; P.scsh = l > 0 ? pn->scsh[l - 1] : nullptr;
        P.w4 = pn->w4[l]; P.bias = pn->bias[l]; P.out = pn->act[l]; P.part = L.ksplit == 1 ? pn->part[l] : nullptr; P.L = L; P.N = N;
        // two pixel blocks per wave in the many-images regime where a layer has pixels to spare (posenet_kernel.h k_pn_conv PB): a fixed
        // function of the regime and the layer, so results stay bit-identical for every batch within a regime
        // (A/B on one box, KITTI odometry sequence at 8 / 12 windows per call: layers 2-5 with two blocks 3 632-3 640 / 3 702-3 710 windows/s,
        // layer 2 only 3 609-3 637 / 3 621-3 626, none 3 504-3 510)
        static const int pb_min_px = getenv("TCSFM_PN_PB_MIN_PIXELS") ? atoi(getenv("TCSFM_PN_PB_MIN_PIXELS")) : 64;        // (measurement hook)
        const int pb = (cfg == 1 && l > 0 && nb >= 2 && L.oh * L.ow >= pb_min_px) ? 2 : 1;
        dim3 grid((L.oh * L.ow + 64 * pb - 1) / (64 * pb), L.cout / (16 * nb), N * L.ksplit);
        if (l == 0) {            // LDS-staged first layer: one workgroup per 64-pixel segment of two output rows
            grid = dim3(((L.oh + 1) / 2) * ((L.ow + 63) / 64), 1, N);
            hipLaunchKernelGGL(k_pn_conv1, grid, dim3(256), 0, h->stream, P);
        } else if (nb == 1) hipLaunchKernelGGL((k_pn_conv<1, false>), grid, dim3(256), 0, h->stream, P);
        else if (nb == 2 && pb == 2) hipLaunchKernelGGL((k_pn_conv<2, false, 2>), grid, dim3(256), 0, h->stream, P);
        else if (nb == 2) hipLaunchKernelGGL((k_pn_conv<2, false>), grid, dim3(256), 0, h->stream, P);
        else if (pb == 2) hipLaunchKernelGGL((k_pn_conv<4, false, 2>), grid, dim3(256), 0, h->stream, P);
        else hipLaunchKernelGGL((k_pn_conv<4, false>), grid, dim3(256), 0, h->stream, P);
        // GroupNorm statistics (+ K-split combination) as their own launch.  Round 3 measured the alternative -- statistics, K-split
        // combination and the head in the convolutions' tails by the last-arriver ticket protocol, 7 launches instead of 15: every
        // convolution became 6-8 us SLOWER (ticket round trips, acquire, serial tail of the last workgroup), 127.5 vs 119 us per
        // evaluation (profiles/r03_posenet_fused_tail_kernel_stats.csv, _timing.jsonl) -- a separate 16 N-workgroup pass is faster.
        PnStatsParams S;
        S.part = P.part; S.tiles = (int)grid.x;
        S.out = pn->act[l]; S.bias = pn->bias[l]; S.gamma = pn->gamma[l]; S.beta = pn->beta[l]; S.scsh = pn->scsh[l];
        S.N = N; S.npix = L.oh * L.ow; S.cout = L.cout; S.ksplit = L.ksplit;
        hipLaunchKernelGGL(k_pn_stats, dim3(N, 16), dim3(256), 0, h->stream, S);
    }
    PnHeadParams Hd;
    Hd.x = pn->act[6]; Hd.scsh = pn->scsh[6]; Hd.w = pn->head_w; Hd.b = pn->head_b; Hd.pose = pose; Hd.stacked = stacked;
    Hd.npix = pn->L[6].oh * pn->L[6].ow; Hd.accumulate = accumulate; Hd.it = it; Hd.iters = iters;
    hipLaunchKernelGGL(k_pn_head, dim3(N), dim3(256), 0, h->stream, Hd);
    HIPCHK(h, hipGetLastError());
    return TCSFM_OK;
}
}  // namespace

int tcsfm_posenet_forward(tcsfm_posenet *pn, int N, const float *imgs, float *pose_out) {
    if (!pn) return TCSFM_E_ARG;
    tcsfm_ctx *h = pn->h;
    if (!pn->loaded) return fail(h, TCSFM_E_ARG, "tcsfm_posenet_forward: no weights loaded");
    if (N < 1 || N > pn->max_images || !imgs || !pose_out) return fail(h, TCSFM_E_ARG, "tcsfm_posenet_forward: bad argument");
    if (int rc_q = drain_queued(h)) return rc_q;
    DeviceGuard dev_guard(h->device);
    const long long hw = (long long)h->H * h->W;
    return pn_run(pn, N, imgs, 6 * hw, imgs + 3 * hw, 6 * hw, 0, 0, pose_out, 0, nullptr, 0, 1);
}

// the coupled loop of train_mono.py:64-80 on context `h` (the handle or one of its lanes; pn->h == h): network, warps, corrections
static int pose_loop(tcsfm_ctx *h, tcsfm_posenet *pn, int num_iter, int B, int S, const float *tgt, const float *srcs, const float *depth_t,
                     const float *depth_s, const float *K, float *poses_out, float *stacked_out, const WinOff *wo) {
    const int N = 2 * B * S;
    int rc;
    tcsfm_opts o; tcsfm_default_opts(&o);
    if ((rc = check_intrinsics(h, &o, K, B))) return rc;
    const long long hw = (long long)h->H * h->W;
    // full_poses = pose_model(cat(tgt | src ; src | tgt)), train_mono.py:54-64 -- the pairs are formed by indexing
    if ((rc = pn_run(pn, N, tgt, 3 * hw, srcs, 3 * hw, B, S, pn->pose, 0, stacked_out, 0, num_iter, wo))) return rc;
    for (int it = 1; it < num_iter; it++) {
        // inverse_warp2(src, d_t, d_s, -full_poses, K) with the next network input (tgt * valid | img_rec) written by the warp
        // itself (train_mono.py:69-76), then full_poses += pose_model(new_imgs) (:77-78)
        InitParams I = init_params(h, &o, N, pn->pose, nullptr, K, 0);
        I.K_mod = B;
        hipLaunchKernelGGL(k_init, dim3((N + 63) / 64), dim3(64), 0, h->stream, I);
        WarpParams W;
        memset(&W, 0, sizeof(W));
        W.src = srcs; W.depth_t = depth_t; W.depth_s = depth_s; W.pc = h->pconst; W.tgt = tgt; W.posenet_in = pn->in_buf;
        W.H = h->H; W.W = h->W; W.win_B = B; W.win_S = S;
        if (wo) W.win_off = *wo;
        hipLaunchKernelGGL(k_warp, dim3((unsigned)((hw + 255) / 256), N), dim3(256), 0, h->stream, W);
        if ((rc = pn_run(pn, N, pn->in_buf, 6 * hw, pn->in_buf + 3 * hw, 6 * hw, 0, 0, pn->pose, 1, stacked_out, it, num_iter))) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(poses_out, pn->pose, (size_t)N * 6 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return TCSFM_OK;
}

int tcsfm_solve_pose_iteratively(tcsfm_handle h, tcsfm_posenet *pn, int num_iter, int B, int S, const float *tgt, const float *srcs,
                                 const float *depth_t, const float *depth_s, const float *K, float *poses_out, float *stacked_out) {
    if (!h || !pn || pn->h != h) return TCSFM_E_ARG;
    if (!pn->loaded) return fail(h, TCSFM_E_ARG, "tcsfm_solve_pose_iteratively: no weights loaded");
    const int N = 2 * B * S;
    if (num_iter < 1 || B < 1 || S < 1 || N > pn->max_images || N > h->max_pairs) return fail(h, TCSFM_E_ARG, "tcsfm_solve_pose_iteratively: sizes out of range");
    if (!tgt || !srcs || !depth_t || !depth_s || !K || !poses_out) return fail(h, TCSFM_E_ARG, "tcsfm_solve_pose_iteratively: NULL argument");
    if (int rc_q = drain_queued(h)) return rc_q;
    DeviceGuard dev_guard(h->device);
    if (int rc_ = pending_error(h)) return rc_;
    return pose_loop(h, pn, num_iter, B, S, tgt, srcs, depth_t, depth_s, K, poses_out, stacked_out, nullptr);
}

void tcsfm_pose_to_matrix(const double pose[6], double T[12]) { tc::pose_to_T(pose, T); }
void tcsfm_matrix_to_pose(const double T[12], double pose[6]) { tc::T_to_pose(T, pose); }
void tcsfm_se3_exp(const double xi[6], double T[12]) { tc::se3_exp(xi, T); }
void tcsfm_se3_log(const double T[12], double xi[6]) { tc::se3_log(T, xi); }
void tcsfm_se3_mul(const double A[12], const double B[12], double C[12]) { tc::se3_mul(A, B, C); }
void tcsfm_se3_inv(const double A[12], double B[12]) { tc::se3_inv(A, B); }

}  // extern "C"
