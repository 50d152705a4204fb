// Frame-parallel training step, the part behind the gradient: Adam on the FLAT parameter buffer (this file) and the direct all-reduce
// over peer pointers (below).  SURVEY.md 8(e): one process per GPU renders its own frame(s); the only exchange is the sum of the
// flat fp32 gradient buffer (951 023 floats at 55 104 Gaussians), after which every rank applies the same optimizer step.
//
// The reference's optimizer is torch.optim.Adam(param_groups, betas=(0.9, 0.999)) with one learning rate per group
// (train.py:263-267, models/model.py:305-327, update_lr train.py:166-175); through torch that is ~10 multi-tensor launches and
// ~100 us of host time per step -- half of a single-frame step (0.2 ms) of this path.  Here it is ONE launch over the flat buffer:
// 16 bytes read and 12 written per parameter, HBM-bound (951 023 parameters: 26.6 MB, ~6 us).
#include <unistd.h>
#include "gom_internal.h"
#include <math.h>
#include <string.h>

namespace {

struct AdamSegs {
    uint32_t begin[GOM_ADAM_MAX_SEGMENTS + 1];   // segment i = [begin[i], begin[i + 1]) of the flat buffer
    float step_size[GOM_ADAM_MAX_SEGMENTS];      // lr_i / (1 - beta1^t_i); with a device step counter: lr_i, corrected in the kernel
    float isb2[GOM_ADAM_MAX_SEGMENTS];           // 1 / sqrt(1 - beta2^t_i); t_i = the segment's OWN step count (a module that joined late, seg_start)
    int32_t start[GOM_ADAM_MAX_SEGMENTS];        // device step counter only: steps that had passed when the segment joined (t_i = t - start_i)
    uint32_t inactive;                           // bit i: segment i has not joined yet -> its elements are left alone, like padding
    int n;
    float omb1, omb2;                            // 1 - beta as torch forms it: in DOUBLE from the decimal the caller wrote, rounded once (one_minus_beta below)
};

// torch computes `1 - beta2` in double from the Python float 0.999 and hands 0.001f to its kernels; this ABI receives beta2 as a FLOAT
// (0.999f = 0.99900001287...), and 1.f - 0.999f = 0.00100004673f is 4.7e-5 away from that.  A float beta only pins 1 - beta to ~6e-8 absolute
// anyway, so the decimal the caller most plausibly wrote (7 digits) is as good a reading as any -- and it is torch's bits for 0.9 / 0.999.
static float one_minus_beta(float beta) { return (float)(1.0 - round((double)beta * 1e7) / 1e7); }

// torch.optim.Adam (no weight decay, no amsgrad, maximize = False), the arithmetic of torch/optim/adam.py::_single_tensor_adam:
//   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g g;  p -= step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// -> false when element e lies outside every segment (padding of the payload: not a parameter, untouched)
__device__ __forceinline__ bool adam_element(const AdamSegs &segs, uint32_t e, float gr, float &pp, float &mm, float &vv, float beta1, float beta2, float eps) {
    float ss = 0.f, inv_sqrt_bc2 = 0.f;
    bool in = false;
#pragma unroll
    for (int s = 0; s < GOM_ADAM_MAX_SEGMENTS; s++)
        if (s < segs.n && e >= segs.begin[s] && e < segs.begin[s + 1]) { ss = segs.step_size[s]; inv_sqrt_bc2 = segs.isb2[s]; in = !((segs.inactive >> s) & 1u); }
    if (!in) return false;
    mm = __fmaf_rn(beta1, mm, __fmul_rn(segs.omb1, gr));              // exp_avg.lerp_(grad, 1 - beta1) up to rounding
    vv = __fmaf_rn(beta2, vv, __fmul_rn(__fmul_rn(segs.omb2, gr), gr));
    const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vv), inv_sqrt_bc2), eps);
    pp = __fsub_rn(pp, __fmul_rn(ss, __fdiv_rn(mm, denom)));
    return true;
}
__global__ void __launch_bounds__(256) k_adam_flat(uint32_t n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, AdamSegs segs, float beta1, float beta2, float eps,
                                                   float grad_scale, long long *__restrict__ step_dev, float lr_decay_steps) {
    // Device-resident step count (step_dev[0] = steps taken so far, step_dev[1] = workgroups of this launch that are done): nothing of
    // the step number is baked into the launch, so the optimizer can sit inside a captured graph.  Every workgroup reads the count
    // before it works; the last one to finish advances it.
    if (step_dev) {
        // Once per workgroup: threads 0 .. GOM_ADAM_MAX_SEGMENTS-1 derive one segment's step size each (three double-precision pow() per
        // THREAD of the launch used to be most of this ~6 us kernel's arithmetic), LDS hands the results to everyone.  Same expressions, same bits.
        __shared__ float s_step[GOM_ADAM_MAX_SEGMENTS], s_isb2[GOM_ADAM_MAX_SEGMENTS];
        if (threadIdx.x < GOM_ADAM_MAX_SEGMENTS) {
            // (the only writer of step_dev[0] is the last workgroup of a launch, behind every workgroup's read: the atomic load states that)
            const double t = (double)(__hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1);
            float base = 0.f;
            int start = 0;
#pragma unroll
            for (int s2 = 0; s2 < GOM_ADAM_MAX_SEGMENTS; s2++) {   // (no dynamic index into the kernel arguments)
                base = (int)threadIdx.x == s2 ? segs.step_size[s2] : base;
                start = (int)threadIdx.x == s2 ? segs.start[s2] : start;
            }
            const double ti = fmax(t - (double)start, 1.0);   // the segment's own step count (bias corrections)
            const double bc1 = 1.0 - pow((double)beta1, ti), bc2 = 1.0 - pow((double)beta2, ti);
            // update_lr (train.py:166-175) runs after the step of iteration n_iters (which counts from 1, train.py:268) with n_iters as its
            // argument: step t uses base * 0.1^((t - 1) / D) -- the ITERATION, not the segment's own count
            const double decay = lr_decay_steps > 0.f ? pow(0.1, (t - 1.0) / (double)lr_decay_steps) : 1.0;
            s_step[threadIdx.x] = (float)((double)base * decay / bc1);
            s_isb2[threadIdx.x] = (float)(1.0 / sqrt(bc2));
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < GOM_ADAM_MAX_SEGMENTS; s2++) { segs.step_size[s2] = s_step[s2]; segs.isb2[s2] = s_isb2[s2]; }
    }
    // 4 consecutive parameters per thread and trip (the buffers come from hipMalloc / torch: 16-byte aligned); the ragged end one by one
    const uint32_t n4 = n >> 2;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n4 + (n & 3u); i += gridDim.x * 256) {
        const bool vec = i < n4;
        const uint32_t e0 = vec ? 4u * i : 4u * n4 + (i - n4);
        const int cnt = vec ? 4 : 1;
        float pp[4], gg[4], mm[4], vv[4];
        if (vec) {
            const float4 a = reinterpret_cast<const float4 *>(p)[i], b = reinterpret_cast<const float4 *>(g)[i];
            const float4 c = reinterpret_cast<const float4 *>(m)[i], d = reinterpret_cast<const float4 *>(v)[i];
            pp[0] = a.x; pp[1] = a.y; pp[2] = a.z; pp[3] = a.w; gg[0] = b.x; gg[1] = b.y; gg[2] = b.z; gg[3] = b.w;
            mm[0] = c.x; mm[1] = c.y; mm[2] = c.z; mm[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
        } else {
            pp[0] = p[e0]; gg[0] = g[e0]; mm[0] = m[e0]; vv[0] = v[e0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (u >= cnt) break;
            adam_element(segs, e0 + (uint32_t)u, gg[u] * grad_scale, pp[u], mm[u], vv[u], beta1, beta2, eps);
        }
        if (vec) {
            reinterpret_cast<float4 *>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
            reinterpret_cast<float4 *>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4 *>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            p[e0] = pp[0]; m[e0] = mm[0]; v[e0] = vv[0];
        }
    }
    if (step_dev) {
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(reinterpret_cast<unsigned long long *>(step_dev + 1), 1ull) == (unsigned long long)gridDim.x - 1ull) {
            step_dev[1] = 0;
            step_dev[0] = step_dev[0] + 1;
        }
    }
}

}  // namespace

// seg_start (may be NULL: every segment has stepped from the beginning): seg_start[i] = optimizer steps that had been taken when segment i joined
// (torch.optim.Adam skips a parameter whose .grad is None and counts ITS steps from its first gradient: the non-rigid and pose-refinement
// MLPs, which Model.forward leaves out until their kick_in_iter, models/model.py:193,200); seg_start[i] < 0 = not joined yet: left alone.
// device_count: the step count lives on the device -- bias corrections are formed in the kernel, step_size carries the bare learning rate.
static int adam_segments(AdamSegs &segs, int64_t n, int32_t n_segments, const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step,
                         double beta1, double beta2, bool device_count) {
    if (n_segments < 1 || n_segments > GOM_ADAM_MAX_SEGMENTS || !seg_begin || !seg_lr) { gom_set_error("Adam: 1..%d segments", GOM_ADAM_MAX_SEGMENTS); return -1; }
    segs.n = n_segments;
    segs.inactive = 0u;
    for (int i = 0; i <= n_segments; i++) {
        if (seg_begin[i] < 0 || seg_begin[i] > n || (i > 0 && seg_begin[i] < seg_begin[i - 1])) { gom_set_error("Adam: segment bounds must ascend inside [0, n]"); return -1; }
        segs.begin[i] = (uint32_t)seg_begin[i];
    }
    for (int i = 0; i < n_segments; i++) {
        const int64_t st = seg_start ? seg_start[i] : 0;
        if (st < 0 || (!device_count && st >= step)) { segs.inactive |= 1u << i; segs.step_size[i] = 0.f; segs.isb2[i] = 0.f; segs.start[i] = 0; continue; }
        if (st > 0x7fffffffLL) { gom_set_error("Adam: seg_start out of range"); return -1; }
        segs.start[i] = (int32_t)st;
        const double ti = (double)(step - st);
        segs.step_size[i] = device_count ? seg_lr[i] : (float)((double)seg_lr[i] / (1.0 - pow(beta1, ti)));
        segs.isb2[i] = device_count ? 0.f : (float)(1.0 / sqrt(1.0 - pow(beta2, ti)));
    }
    return 0;
}

extern "C" int gom_adam_flat(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                             const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps,
                             float grad_scale, void *stream) {
    return gom_adam_flat_graphable(n, params, grads, exp_avg, exp_avg_sq, n_segments, seg_begin, seg_lr, seg_start, step, nullptr, 0.f, beta1, beta2, eps, grad_scale,
                                   stream);
}

extern "C" int gom_adam_flat_graphable(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                                       const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, int64_t *step_device,
                                       float lr_decay_steps, float beta1, float beta2, float eps, float grad_scale, void *stream) {
    if (n < 0 || n > 0xffffffffLL) { gom_set_error("gom_adam_flat: bad size"); return -1; }
    if (n_segments < 1 || n_segments > GOM_ADAM_MAX_SEGMENTS || !seg_begin || !seg_lr) { gom_set_error("gom_adam_flat: 1..%d segments", GOM_ADAM_MAX_SEGMENTS); return -1; }
    if (!step_device && step < 1) { gom_set_error("gom_adam_flat: step counts from 1"); return -1; }
    if (step_device) step = 1;   // (host-side corrections unused: the kernel derives them from the device count)
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq) { gom_set_error("gom_adam_flat: null pointer"); return -1; }
    AdamSegs segs{};
    if (int rc = adam_segments(segs, n, n_segments, seg_begin, seg_lr, seg_start, step, beta1, beta2, step_device != nullptr)) return rc;
    segs.omb1 = one_minus_beta(beta1); segs.omb2 = one_minus_beta(beta2);
    const uint32_t work = (uint32_t)(n >> 2) + (uint32_t)(n & 3);
    const unsigned blocks = (unsigned)((work + 255) / 256);
    hipLaunchKernelGGL(k_adam_flat, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream, (uint32_t)n, params, grads, exp_avg, exp_avg_sq, segs,
                       beta1, beta2, eps, grad_scale, reinterpret_cast<long long *>(step_device), lr_decay_steps);
    GOM_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------------------------
// The same step over a LIST of tensors (gom_adam_multi): the reference's optimizer as the reference builds it -- torch.optim.Adam over
// Model.get_param_groups() (train.py:263-267; parameters, gradients and moments are separate allocations owned by torch) -- is ~35
// multi_tensor_apply launches and 0.58 ms per iteration of the drop-in Model's 4.4 ms (profiles/r03_model_train_iteration_kernel_stats.csv).
// One launch here: the tensors' pointers, sizes and step sizes travel in the kernel arguments, a workgroup owns 1 024 consecutive
// elements of one tensor.  gomavatar_amd.optim.GomAdam (a torch.optim.Optimizer) sits on top.
namespace {
struct AdamMulti {
    float *p[GOM_ADAM_MULTI_MAX];
    const float *g[GOM_ADAM_MULTI_MAX];
    float *m[GOM_ADAM_MULTI_MAX], *v[GOM_ADAM_MULTI_MAX];
    uint32_t n[GOM_ADAM_MULTI_MAX];
    uint32_t blk0[GOM_ADAM_MULTI_MAX + 1];   // first workgroup of tensor i
    float lr[GOM_ADAM_MULTI_MAX];
    int nt;
};
__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti a, float beta1, float beta2, float omb1, float omb2, float eps, float inv_bc1, float inv_sqrt_bc2,
                                                    const long long *__restrict__ step_dev) {
    if (step_dev) {   // device-resident step count (captured graphs): step_dev[0] = steps taken so far; the caller advances it (gom_adam_multi enqueues k_adam_multi_tick)
        __shared__ float s_c[2];
        if (threadIdx.x == 0) {
            const double t = (double)(__hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1);
            s_c[0] = (float)(1.0 / (1.0 - pow((double)beta1, t)));
            s_c[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
        }
        __syncthreads();
        inv_bc1 = s_c[0]; inv_sqrt_bc2 = s_c[1];
    }
    int t = 0;
#pragma unroll
    for (int i = 1; i < GOM_ADAM_MULTI_MAX; i++) t = (i < a.nt && blockIdx.x >= a.blk0[i]) ? i : t;
    float *p = nullptr, *m = nullptr, *v = nullptr;
    const float *g = nullptr;
    uint32_t n = 0, b0 = 0;
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < GOM_ADAM_MULTI_MAX; i++)   // (no dynamic index into the kernel arguments)
        if (i == t) { p = a.p[i]; g = a.g[i]; m = a.m[i]; v = a.v[i]; n = a.n[i]; b0 = a.blk0[i]; lr = a.lr[i]; }
    const float ss = __fmul_rn(lr, inv_bc1);   // torch: step_size = lr / bias_correction1
    const uint32_t base = (blockIdx.x - b0) * 1024u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t e = base + 256u * (uint32_t)k;
        if (e < n) {
            const float gr = g[e];
            float mm = m[e], vv = v[e], pp = p[e];
            // (omb = 1 - beta rounded from DOUBLE, as torch passes `1 - beta2` to lerp_ / addcmul_: 1.f - 0.999f is 4.7e-5 away from 0.001f)
            mm = __fmaf_rn(beta1, mm, __fmul_rn(omb1, gr));
            vv = __fmaf_rn(beta2, vv, __fmul_rn(__fmul_rn(omb2, gr), gr));
            const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vv), inv_sqrt_bc2), eps);
            pp = __fsub_rn(pp, __fmul_rn(ss, __fdiv_rn(mm, denom)));
            p[e] = pp; m[e] = mm; v[e] = vv;
        }
    }
}
__global__ void k_adam_multi_tick(long long *step_dev) { step_dev[0] = step_dev[0] + 1; }
}  // namespace

extern "C" int gom_adam_multi(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel,
                              const float *lr, int64_t step, int64_t *step_device, double beta1, double beta2, double eps, void *stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr))) { gom_set_error("gom_adam_multi: null argument"); return -1; }
    if (!step_device && step < 1) { gom_set_error("gom_adam_multi: step counts from 1"); return -1; }
    const double bc1 = 1.0 - pow(beta1, (double)(step_device ? 1 : step)), bc2 = 1.0 - pow(beta2, (double)(step_device ? 1 : step));
    for (int t0 = 0; t0 < n_tensors; t0 += GOM_ADAM_MULTI_MAX) {   // (more tensors than one launch carries: several launches)
        AdamMulti a{};
        uint32_t blocks = 0;
        for (int i = t0; i < n_tensors && i < t0 + GOM_ADAM_MULTI_MAX; i++) {
            if (numel[i] < 0 || numel[i] > 0xffffffffLL) { gom_set_error("gom_adam_multi: bad tensor size"); return -1; }
            if (numel[i] == 0) continue;
            if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) { gom_set_error("gom_adam_multi: null tensor %d", i); return -1; }
            const int k = a.nt++;
            a.p[k] = params[i]; a.g[k] = grads[i]; a.m[k] = exp_avg[i]; a.v[k] = exp_avg_sq[i]; a.n[k] = (uint32_t)numel[i]; a.lr[k] = lr[i];
            a.blk0[k] = blocks;
            blocks += (uint32_t)((numel[i] + 1023) / 1024);
        }
        a.blk0[a.nt] = blocks;
        if (!blocks) continue;
        hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                           (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), reinterpret_cast<const long long *>(step_device));
        GOM_LAUNCH_CHECK();
    }
    if (step_device) {
        hipLaunchKernelGGL(k_adam_multi_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<long long *>(step_device));
        GOM_LAUNCH_CHECK();
    }
    return 0;
}

// =====================================================================================================================================
// Direct all-reduce of the flat gradient buffer over PEER POINTERS: every rank's buffer lives in a region the other
// ranks have mapped (hipIpc handles exchanged once over the process group), and the sum is formed by two kernels per rank, no library
// collective:
//   k_peer_reduce_scatter   rank r raises "my gradient is ready" in every peer's flag row, waits for the peers' flags, sums ITS slice
//                           (n / world elements) of all buffers in RANK ORDER, scales it and leaves it in its own region; its last
//                           workgroup raises "slice r is reduced" at every peer;
//   k_peer_all_gather       waits for those flags and copies every rank's reduced slice.
// Every element is summed by exactly one rank in a fixed order, so all ranks end with the same bits -- bitwise equal to
// ((g_0 + g_1) + g_2) + ... scaled.  The payload (3.8 MB at 55 104 Gaussians) is latency-bound on xGMI: a two-shot exchange moves
// 2 x (world - 1) / world of it per rank over 7 point-to-point links in parallel, where a ring serialises 2 (world - 1) steps.
// The kernels are instantiated for world = 2, 4, 8 (and a generic one): the loads of ALL peers' copies of an element are issued before the
// first (rank-ordered) add, so an element costs one xGMI round trip, not world - 1 of them one behind the other.
// Flags are epoch counters (no reset, no ABA) written with system-scope release stores and polled with system-scope acquire loads; the
// region is FINE-GRAINED memory (coherent across devices without a kernel boundary -- the flags are polled inside a running kernel; creation
// fails where it cannot be had, and the caller falls back to the library collective).
// Failure is loud: a wait gives up after `timeout_s` of wall clock (default 30 s: a peer inside a checkpoint write, an evaluation pass or
// a graph capture is late, not gone), sets the status word -- mirrored into pinned host memory, which gom_peer_reduce_poll reads without
// a device synchronisation, so the host layer checks it every step -- and the rank that timed out raises NO "reduced" flag: its peers'
// all-gathers then time out as well and every rank reports the failure instead of some of them stepping on unreduced data.
// gom_peer_reduce_reset (collective: between two barriers of the group) clears the condition.
// Reuse: a rank overwrites its gradient only after its own all-gather kernel, which has seen every peer's "reduced" flag (= the peer is
// done reading it); it overwrites its reduced slice only in the next scatter kernel, behind the peers' next "ready" flags (= their
// previous all-gather has completed).
//
// ZeRO-1 variant (gom_peer_reduce_run_zero1; SURVEY.md 8(e) "or"): the rank that reduced a slice also applies the Adam step of THAT slice
// -- moments are only ever touched for the own slice -- and leaves the updated PARAMETERS in its region; the second kernel gathers
// parameters instead of gradients.  Same element arithmetic, every element computed once by its owner instead of world times: the
// replicas end with the bits gom_peer_reduce_run_adam gives them.
struct GomPeerReduce {
    int rank = 0, world = 1;
    int64_t n = 0;
    size_t bytes = 0;
    unsigned char *local = nullptr;                 // [grad n][reduced n][flags]
    unsigned char *peer[GOM_PEER_MAX_RANKS] = {};    // mapped regions, own entry = local
    bool opened[GOM_PEER_MAX_RANKS] = {};
    uint32_t epoch = 0;
    uint32_t *status = nullptr;                      // device word: 1 = a wait timed out (sticky until gom_peer_reduce_reset)
    uint32_t *done_ctr = nullptr;                    // workgroups of the scatter kernel that have finished
    uint32_t *status_host = nullptr;                 // pinned host mirror of the status word (written by the kernel that times out)
    uint32_t *status_host_dev = nullptr;             // its device address
    unsigned long long timeout_ticks = 3000000000ull;   // 30 s of the 100 MHz wall clock
};

namespace {
constexpr int kPeerFlagStride = 16;   // uint32 per flag slot (64 bytes: one line per writer)
struct PeerPtrs { unsigned char *p[GOM_PEER_MAX_RANKS]; };
struct PeerStatus { uint32_t *dev, *host; unsigned long long timeout_ticks; };

__device__ __forceinline__ uint32_t *peer_flags(unsigned char *region, int64_t n, int which) {   // which: 0 = ready, 1 = reduced
    return reinterpret_cast<uint32_t *>(region + 2 * (size_t)n * sizeof(float)) + (size_t)which * GOM_PEER_MAX_RANKS * kPeerFlagStride;
}
// true when the flag reached `epoch`; gives up (status = 1, on the device and in the host mirror) after the time limit so that a missing
// peer can never hang the GPU, and at once when another wait of this rank has already given up
__device__ __forceinline__ bool peer_wait(const uint32_t *flag, uint32_t epoch, const PeerStatus &st) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) >= 0) return true;
        __builtin_amdgcn_s_sleep(32);
        if (__hip_atomic_load(st.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (wall_clock64() - t0 > st.timeout_ticks) break;
    }
    __hip_atomic_store(st.dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(st.host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

// W = world size known at compile time (2, 4, 8) or 0: generic.  ZERO1: apply Adam to the slice and publish the PARAMETERS.
template <int W, bool ZERO1>
__global__ void __launch_bounds__(256) k_peer_reduce_scatter(PeerPtrs pp, int rank, int world_rt, int64_t n, uint32_t epoch, float scale, PeerStatus st,
                                                             uint32_t *done_ctr, float *__restrict__ prm, float *__restrict__ m, float *__restrict__ v,
                                                             AdamSegs segs, float beta1, float beta2, float eps, float inv_sqrt_bc2) {
    __shared__ int s_ok;
    const int world = W ? W : world_rt;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)world)   // "my gradient of this epoch is complete" (the kernels that wrote it precede this one on the stream)
        __hip_atomic_store(peer_flags(pp.p[threadIdx.x], n, 0) + rank * kPeerFlagStride, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        bool ok = true;
        for (int p = 0; p < world && ok; p++) ok = peer_wait(peer_flags(pp.p[rank], n, 0) + p * kPeerFlagStride, epoch, st);
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (the waits were thread 0's: the other threads' loads must not be older than the flags)
    const bool ok = s_ok != 0;
    if (ok) {
        const int64_t n4 = n / 4, per = (n4 + world - 1) / world, lo = per * rank, hi = lo + per < n4 ? lo + per : n4;   // float4 units of my slice
        float4 *dst = reinterpret_cast<float4 *>(pp.p[rank] + (size_t)n * sizeof(float));
        const int64_t pend = ZERO1 ? (int64_t)segs.begin[segs.n] : 0;   // the parameter buffers end where the last segment ends; the payload may be padded beyond
        for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
            float4 a;
            if (W) {   // every peer's copy in flight before the first add; the adds in rank order
                float4 b[W ? W : 1];
#pragma unroll
                for (int p = 0; p < W; p++) b[p] = reinterpret_cast<const float4 *>(pp.p[p])[i];
                a = b[0];
#pragma unroll
                for (int p = 1; p < W; p++) { a.x += b[p].x; a.y += b[p].y; a.z += b[p].z; a.w += b[p].w; }
            } else {
                a = reinterpret_cast<const float4 *>(pp.p[0])[i];
                for (int p = 1; p < world; p++) {
                    const float4 b = reinterpret_cast<const float4 *>(pp.p[p])[i];
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
            }
            a = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
            if (ZERO1) {   // my slice's optimizer step; what the peers gather is the updated parameters
                const uint32_t e = (uint32_t)(4 * i);
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * i + 3 < pend) {
                    float4 c = reinterpret_cast<float4 *>(m)[i], d = reinterpret_cast<float4 *>(v)[i];
                    q = reinterpret_cast<float4 *>(prm)[i];
                    adam_element(segs, e, a.x, q.x, c.x, d.x, beta1, beta2, eps);
                    adam_element(segs, e + 1, a.y, q.y, c.y, d.y, beta1, beta2, eps);
                    adam_element(segs, e + 2, a.z, q.z, c.z, d.z, beta1, beta2, eps);
                    adam_element(segs, e + 3, a.w, q.w, c.w, d.w, beta1, beta2, eps);
                    reinterpret_cast<float4 *>(prm)[i] = q; reinterpret_cast<float4 *>(m)[i] = c; reinterpret_cast<float4 *>(v)[i] = d;
                } else {
                    const float gq[4] = {a.x, a.y, a.z, a.w};
                    float qq[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int u = 0; u < 4; u++)
                        if (4 * i + u < pend) { adam_element(segs, e + (uint32_t)u, gq[u], prm[4 * i + u], m[4 * i + u], v[4 * i + u], beta1, beta2, eps); qq[u] = prm[4 * i + u]; }
                    q = make_float4(qq[0], qq[1], qq[2], qq[3]);
                }
                a = q;
            }
            dst[i] = a;
        }
        if (rank == world - 1) {   // the ragged end (n not a multiple of 4) belongs to the last rank
            for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
                float a = reinterpret_cast<const float *>(pp.p[0])[i];
                for (int p = 1; p < world; p++) a += reinterpret_cast<const float *>(pp.p[p])[i];
                a *= scale;
                if (ZERO1) {
                    if (i < pend) { adam_element(segs, (uint32_t)i, a, prm[i], m[i], v[i], beta1, beta2, eps); a = prm[i]; }
                    else a = 0.f;
                }
                reinterpret_cast<float *>(pp.p[rank] + (size_t)n * sizeof(float))[i] = a;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        if (atomicAdd(done_ctr, 1u) == gridDim.x - 1) {   // the last workgroup: my slice is complete everywhere -> tell the peers
            __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (pairs with the other workgroups' release fences in front of their increments)
            *done_ctr = 0;
            // A rank whose wait gave up has summed nothing: it raises NO flag, its peers' all-gathers give up in turn and every rank reports.
            if (__hip_atomic_load(st.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                for (int p = 0; p < world; p++)
                    __hip_atomic_store(peer_flags(pp.p[p], n, 1) + rank * kPeerFlagStride, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// MODE 0: copy every rank's reduced slice to `out`.  MODE 1: the optimizer inside the gather -- instead of copying a rank's reduced slice and
// running Adam over the copy afterwards, every rank applies the Adam step of ITS replica of the parameters straight from the slice as it reads
// it (one launch and one pass over the gradient less: ~6 us of kernel and the ~9 us that follow a plain launch, of a 0.2 ms single-frame
// step).  MODE 2 (ZeRO-1): the slices hold updated PARAMETERS; the own slice is already in place, the others are copied into `p`.
template <int MODE>
__global__ void __launch_bounds__(256) k_peer_all_gather(PeerPtrs pp, int rank, int world, int64_t n, uint32_t epoch, PeerStatus st, float *__restrict__ p,
                                                         float *__restrict__ m, float *__restrict__ v, AdamSegs segs, float beta1, float beta2, float eps,
                                                         float inv_sqrt_bc2, float *__restrict__ out) {
    __shared__ int s_ok;
    const int64_t n4 = n / 4, per = (n4 + world - 1) / world;
    const int64_t pend = MODE ? (int64_t)segs.begin[segs.n] : n;   // the parameter buffers end where the last segment ends; the payload may be padded beyond
    for (int src = 0; src < world; src++) {   // start with my own slice (ready first), then the others in ring order
        const int q = (rank + src) % world;
        if (threadIdx.x == 0) s_ok = peer_wait(peer_flags(pp.p[rank], n, 1) + q * kPeerFlagStride, epoch, st) ? 1 : 0;
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (s_ok && !(MODE == 2 && q == rank)) {
            const int64_t lo = per * q, hi = lo + per < n4 ? lo + per : n4;
            const float4 *srcp = reinterpret_cast<const float4 *>(pp.p[q] + (size_t)n * sizeof(float));
            for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
                const float4 g4 = srcp[i];
                if (MODE == 0) { reinterpret_cast<float4 *>(out)[i] = g4; continue; }
                if (MODE == 1 && out) reinterpret_cast<float4 *>(out)[i] = g4;
                const uint32_t e = (uint32_t)(4 * i);
                if (4 * i + 3 < pend) {
                    if (MODE == 2) { reinterpret_cast<float4 *>(p)[i] = g4; continue; }
                    float4 a = reinterpret_cast<float4 *>(p)[i], c = reinterpret_cast<float4 *>(m)[i], d = reinterpret_cast<float4 *>(v)[i];
                    adam_element(segs, e, g4.x, a.x, c.x, d.x, beta1, beta2, eps);
                    adam_element(segs, e + 1, g4.y, a.y, c.y, d.y, beta1, beta2, eps);
                    adam_element(segs, e + 2, g4.z, a.z, c.z, d.z, beta1, beta2, eps);
                    adam_element(segs, e + 3, g4.w, a.w, c.w, d.w, beta1, beta2, eps);
                    reinterpret_cast<float4 *>(p)[i] = a; reinterpret_cast<float4 *>(m)[i] = c; reinterpret_cast<float4 *>(v)[i] = d;
                } else {
                    const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
                    for (int u = 0; u < 4; u++)
                        if (4 * i + u < pend) {
                            if (MODE == 2) p[4 * i + u] = gq[u];
                            else adam_element(segs, e + (uint32_t)u, gq[u], p[4 * i + u], m[4 * i + u], v[4 * i + u], beta1, beta2, eps);
                        }
                }
            }
            if (q == world - 1)
                for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
                    const float g1 = reinterpret_cast<const float *>(pp.p[q] + (size_t)n * sizeof(float))[i];
                    if (MODE == 0) { out[i] = g1; continue; }
                    if (MODE == 1 && out) out[i] = g1;
                    if (i < pend) {
                        if (MODE == 2) p[i] = g1;
                        else adam_element(segs, (uint32_t)i, g1, p[i], m[i], v[i], beta1, beta2, eps);
                    }
                }
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" GomPeerReduce *gom_peer_reduce_create(int32_t rank, int32_t world, int64_t n_floats) {
    if (world < 1 || world > GOM_PEER_MAX_RANKS || rank < 0 || rank >= world || n_floats <= 0) { gom_set_error("gom_peer_reduce_create: bad arguments"); return nullptr; }
    GomPeerReduce *h = new GomPeerReduce();
    h->rank = rank; h->world = world; h->n = n_floats;
    h->bytes = 2 * (size_t)n_floats * sizeof(float) + 2 * GOM_PEER_MAX_RANKS * kPeerFlagStride * sizeof(uint32_t);
    h->bytes = (h->bytes + 4095) & ~(size_t)4095;
    // Fine-grained: coherent across devices without a kernel boundary (the flags are polled, and the peers' data read, INSIDE a running kernel).
    // Plain (coarse-grained) memory would let the polls see stale lines: no fallback -- the caller uses the library collective instead.
    if (hipExtMallocWithFlags((void **)&h->local, h->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        gom_set_error("gom_peer_reduce_create: fine-grained device memory is not available on this device (the peer exchange needs it; use the collective)");
        delete h;
        return nullptr;
    }
    if (hipMemset(h->local, 0, h->bytes) != hipSuccess || hipMalloc((void **)&h->status, 2 * sizeof(uint32_t)) != hipSuccess || hipMemset(h->status, 0, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void **)&h->status_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void **)&h->status_host_dev, h->status_host, 0) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        gom_set_error("gom_peer_reduce_create: initialisation failed");
        if (h->local) (void)hipFree(h->local);
        if (h->status) (void)hipFree(h->status);
        if (h->status_host) (void)hipHostFree(h->status_host);
        delete h;
        return nullptr;
    }
    *h->status_host = 0;
    h->done_ctr = h->status + 1;
    h->peer[rank] = h->local;
    return h;
}

extern "C" int gom_peer_reduce_handle(GomPeerReduce *h, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!h || !handle64) { gom_set_error("gom_peer_reduce_handle: null argument"); return -1; }
    // (seen once in ~25 multi-process runs on one device: `invalid argument` while several processes export their regions at the same moment;
    //  the same call succeeds a few milliseconds later)
    // Round 6: with EIGHT processes on one device the failure was seen to PERSIST for a region (six attempts over 1.3 s, one run in six): after three
    // attempts the region is allocated anew (the old one is freed only behind the new allocation, so the new one lies elsewhere) -- nobody holds its
    // address yet (peers map it from this handle, the caller asks for gom_peer_reduce_buffer after the exchange of the handles).
    hipError_t e = hipSuccess;
    for (int round = 0; round < 4; round++) {
        for (int attempt = 0; attempt < 3; attempt++) {
            e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle64), h->local);
            if (e == hipSuccess) return 0;
            (void)hipGetLastError();
            usleep(20000 << attempt);
        }
        unsigned char *fresh = nullptr;
        if (hipExtMallocWithFlags((void **)&fresh, h->bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(fresh, 0, h->bytes) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            if (fresh) (void)hipFree(fresh);
            break;
        }
        (void)hipFree(h->local);
        h->local = fresh;
        h->peer[h->rank] = h->local;
    }
    gom_set_error("hipIpcGetMemHandle failed after 12 attempts on 4 regions: %s", hipGetErrorString(e));
    return -2;
}

extern "C" int gom_peer_reduce_connect(GomPeerReduce *h, const void *handles) {
    if (!h || !handles) { gom_set_error("gom_peer_reduce_connect: null argument"); return -1; }
    for (int p = 0; p < h->world; p++) {
        if (p == h->rank || h->opened[p]) continue;
        hipIpcMemHandle_t hd;
        memcpy(&hd, (const unsigned char *)handles + 64 * (size_t)p, 64);
        hipError_t e = hipSuccess;
        for (int attempt = 0; attempt < 6; attempt++) {   // (same transient as in gom_peer_reduce_handle)
            e = hipIpcOpenMemHandle((void **)&h->peer[p], hd, hipIpcMemLazyEnablePeerAccess);
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            usleep(20000 << attempt);
        }
        if (e != hipSuccess) { gom_set_error("hipIpcOpenMemHandle (rank %d) failed after 6 attempts: %s", p, hipGetErrorString(e)); return -2; }
        h->opened[p] = true;
    }
    return 0;
}

extern "C" float *gom_peer_reduce_buffer(GomPeerReduce *h) { return h ? reinterpret_cast<float *>(h->local) : nullptr; }

extern "C" int gom_peer_reduce_set_timeout(GomPeerReduce *h, double seconds) {
    if (!h || !(seconds > 0.0) || seconds > 3600.0) { gom_set_error("gom_peer_reduce_set_timeout: 0 < seconds <= 3600"); return -1; }
    h->timeout_ticks = (unsigned long long)(seconds * 1e8);   // wall_clock64 counts at 100 MHz
    return 0;
}

static bool peer_ready(GomPeerReduce *h, const char *who) {
    if (!h) { gom_set_error("%s: null handle", who); return false; }
    for (int p = 0; p < h->world; p++)
        if (!h->peer[p]) { gom_set_error("%s: rank %d is not connected", who, p); return false; }
    if (*reinterpret_cast<volatile uint32_t *>(h->status_host)) { gom_set_error("%s: an earlier exchange timed out (a peer did not answer); gom_peer_reduce_reset on every rank first", who); return false; }
    return true;
}

// scatter kernel of the right instantiation
template <bool ZERO1>
static void launch_scatter(GomPeerReduce *h, const PeerPtrs &pp, float scale, const PeerStatus &st, float *prm, float *m, float *v, const AdamSegs &segs, float beta1,
                           float beta2, float eps, float isb2, hipStream_t stream) {
    // small resident grids: every workgroup polls flags, so all of them must fit on the chip next to whatever else runs
    const dim3 grid(64), block(256);
#define GOM_PS(WW) hipLaunchKernelGGL((k_peer_reduce_scatter<WW, ZERO1>), grid, block, 0, stream, pp, h->rank, h->world, h->n, h->epoch, scale, st, h->done_ctr, prm, m, v, segs, beta1, beta2, eps, isb2)
    switch (h->world) {
        case 2: GOM_PS(2); break;
        case 4: GOM_PS(4); break;
        case 8: GOM_PS(8); break;
        default: GOM_PS(0); break;
    }
#undef GOM_PS
}

extern "C" int gom_peer_reduce_run(GomPeerReduce *h, float *out, float scale, void *stream) {
    if (!out) { gom_set_error("gom_peer_reduce_run: null argument"); return -1; }
    if (!peer_ready(h, "gom_peer_reduce_run")) return -1;
    h->epoch++;
    PeerPtrs pp{};
    for (int p = 0; p < h->world; p++) pp.p[p] = h->peer[p];
    const PeerStatus st{h->status, h->status_host_dev, h->timeout_ticks};
    const AdamSegs none{};
    launch_scatter<false>(h, pp, scale, st, nullptr, nullptr, nullptr, none, 0.f, 0.f, 0.f, 0.f, (hipStream_t)stream);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_peer_all_gather<0>, dim3(64), dim3(256), 0, (hipStream_t)stream, pp, h->rank, h->world, h->n, h->epoch, st, nullptr, nullptr, nullptr, none, 0.f, 0.f, 0.f,
                       0.f, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

static int peer_adam_common(GomPeerReduce *h, const char *who, float *params, float *exp_avg, float *exp_avg_sq, int32_t n_segments, const int64_t *seg_begin,
                            const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, AdamSegs &segs) {
    if (!params || !exp_avg || !exp_avg_sq) { gom_set_error("%s: null argument", who); return -1; }
    if (step < 1) { gom_set_error("%s: step counts from 1", who); return -1; }
    if (!peer_ready(h, who)) return -1;
    if (int rc = adam_segments(segs, h->n, n_segments, seg_begin, seg_lr, seg_start, step, beta1, beta2, false)) return rc;
    segs.omb1 = one_minus_beta(beta1); segs.omb2 = one_minus_beta(beta2);
    return 0;
}

extern "C" int gom_peer_reduce_run_adam(GomPeerReduce *h, float scale, float *out, float *params, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                                        const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps, void *stream) {
    AdamSegs segs{};
    const float isb2 = 0.f;   // (per segment, inside `segs`)
    if (int rc = peer_adam_common(h, "gom_peer_reduce_run_adam", params, exp_avg, exp_avg_sq, n_segments, seg_begin, seg_lr, seg_start, step, beta1, beta2, segs)) return rc;
    h->epoch++;
    PeerPtrs pp{};
    for (int p = 0; p < h->world; p++) pp.p[p] = h->peer[p];
    const PeerStatus st{h->status, h->status_host_dev, h->timeout_ticks};
    launch_scatter<false>(h, pp, scale, st, nullptr, nullptr, nullptr, segs, 0.f, 0.f, 0.f, 0.f, (hipStream_t)stream);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_peer_all_gather<1>, dim3(64), dim3(256), 0, (hipStream_t)stream, pp, h->rank, h->world, h->n, h->epoch, st, params, exp_avg, exp_avg_sq, segs, beta1,
                       beta2, eps, isb2, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_peer_reduce_run_zero1(GomPeerReduce *h, float scale, float *params, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                                         const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps, void *stream) {
    AdamSegs segs{};
    const float isb2 = 0.f;   // (per segment, inside `segs`)
    if (int rc = peer_adam_common(h, "gom_peer_reduce_run_zero1", params, exp_avg, exp_avg_sq, n_segments, seg_begin, seg_lr, seg_start, step, beta1, beta2, segs)) return rc;
    h->epoch++;
    PeerPtrs pp{};
    for (int p = 0; p < h->world; p++) pp.p[p] = h->peer[p];
    const PeerStatus st{h->status, h->status_host_dev, h->timeout_ticks};
    launch_scatter<true>(h, pp, scale, st, params, exp_avg, exp_avg_sq, segs, beta1, beta2, eps, isb2, (hipStream_t)stream);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_peer_all_gather<2>, dim3(64), dim3(256), 0, (hipStream_t)stream, pp, h->rank, h->world, h->n, h->epoch, st, params, nullptr, nullptr, segs, 0.f, 0.f, 0.f,
                       0.f, nullptr);
    GOM_LAUNCH_CHECK();
    return 0;
}

// Non-blocking: the pinned host mirror of the status word (a kernel that gives up writes it through the mapping).  0 = no wait has timed out
// SO FAR -- a step's own kernels may still be running; the host layer calls this every step and so learns of a failure one step late at most.
extern "C" int gom_peer_reduce_poll(GomPeerReduce *h) {
    if (!h) return -1;
    const uint32_t s = *reinterpret_cast<volatile uint32_t *>(h->status_host);
    if (s) gom_set_error("gom_peer_reduce: a peer did not answer within the wait limit");
    return (int)s;
}

extern "C" int gom_peer_reduce_status(GomPeerReduce *h) {   // host-side check (synchronises the device): 0 = every wait was answered
    if (!h) return -1;
    uint32_t s = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&s, h->status, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) { gom_set_error("gom_peer_reduce_status: device error"); return -2; }
    if (s) gom_set_error("gom_peer_reduce: a peer did not answer within the wait limit");
    return (int)s;
}

// After a timeout.  COLLECTIVE in spirit: the caller brackets it with two barriers of the process group (every rank idle before anybody
// clears; everybody cleared before anybody starts an exchange).  Flags are epoch counters: this rank's epoch moves to `epoch` (the maximum
// over the ranks, agreed by the caller) so that stale flags of the failed exchange can never satisfy a later wait.
extern "C" int gom_peer_reduce_reset(GomPeerReduce *h, uint32_t epoch) {
    if (!h) { gom_set_error("gom_peer_reduce_reset: null handle"); return -1; }
    GOM_HIP_CHECK(hipDeviceSynchronize());
    GOM_HIP_CHECK(hipMemset(h->status, 0, 2 * sizeof(uint32_t)));
    GOM_HIP_CHECK(hipDeviceSynchronize());
    *h->status_host = 0;
    if ((int32_t)(epoch - h->epoch) > 0) h->epoch = epoch;
    return 0;
}
extern "C" uint32_t gom_peer_reduce_epoch(GomPeerReduce *h) { return h ? h->epoch : 0u; }

// The caller has synchronised its device AND passed a barrier of the process group: no peer kernel may still be reading this region.
extern "C" void gom_peer_reduce_destroy(GomPeerReduce *h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < h->world; p++)
        if (h->opened[p] && h->peer[p]) (void)hipIpcCloseMemHandle(h->peer[p]);
    if (h->local) (void)hipFree(h->local);
    if (h->status) (void)hipFree(h->status);
    if (h->status_host) (void)hipHostFree(h->status_host);
    delete h;
}
