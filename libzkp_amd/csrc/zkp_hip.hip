// libzkp_hip: kernels + C ABI (include/libzkp_hip.h) of the MI355X Bulletproofs prover (range / threshold / consistency
// framings, 8- to 64-bit proofs) and, through the .inc files at the end, the host side of the Groth16, STARK, verification
// and mixed-batch entry points.  gfx950 only.  One lane = one proof for every scalar / transcript step; the dominant
// kernel (k_msm_gather, msm_kernel.h) walks fixed-base window tables sized for HBM, one gathered entry per lane and step.
#include <hip/hip_runtime.h>
#include <mutex>
#include <memory>
#include <algorithm>
#include <map>
#include <array>
#include <type_traits>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <functional>
#include <deque>
#include <new>
#include <exception>
#include <cstring>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <sys/random.h>
#include "bp_layout.h"
#include "g16_steps.h"
#include "sha256_host.h"
#include "g16_circuit.h"
#include "msm_kernel.h"
#include "g16_launch.h"
#include "g16_verify_launch.h"
#include "stark_launch.h"
#include "bpv_launch.h"
#include "edg_launch.h"
#include "../../include/libzkp_hip.h"

// ================================================================================================ kernels

__global__ void __launch_bounds__(TB) k_build_range(JobBuf J, uint32_t n, const uint64_t* value, const uint64_t* mn, const uint64_t* mx, uint32_t lg,
                                                    uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status) {
    ZKP_RAISE_PRIO();
    const uint32_t op = blockIdx.x * TB + threadIdx.x;
    if (op < n) step_build_range(J, op, value, mn, mx, lg, out, stride, out_len, status);
}
__global__ void __launch_bounds__(TB) k_ctask(CtView T) {
    ZKP_RAISE_PRIO();
    const uint32_t c = blockIdx.x * TB + threadIdx.x;
    if (c < T.C) step_ctask(T, c);
}
__global__ void __launch_bounds__(TB) k_tape(BpView V) {
    ZKP_RAISE_PRIO();
    const uint32_t job = blockIdx.x * TB + threadIdx.x;
    if (job < V.M) step_tape(V, blockIdx.y, job);
}
__global__ void __launch_bounds__(TB, ZKP_LAT_WAVES) k_poly(BpView V) {
    ZKP_RAISE_PRIO();
    const uint32_t job = blockIdx.x * TB + threadIdx.x;
    if (job < V.M) step_poly(V, blockIdx.y, job);
}
// eight lanes per job (block = 8 jobs x 8 parts): each adds every 8th entry, a 3-level tree through LDS joins the parts
__device__ __forceinline__ ScTriple triple_tree8(ScTriple t, uint32_t* lds) {
    const uint32_t lane = threadIdx.x, grp = lane >> 3;
    for (uint32_t stride = 4; stride >= 1; stride >>= 1) {
        if (grp >= stride && grp < 2 * stride) {
            ZKP_UNROLL for (int k = 0; k < 8; k++) { lds[k * TW + lane] = t.a.v[k]; lds[(8 + k) * TW + lane] = t.b.v[k]; lds[(16 + k) * TW + lane] = t.c.v[k]; }
        }
        __syncthreads();
        if (grp < stride) {
            ScTriple o; const uint32_t src = lane + stride * 8;
            ZKP_UNROLL for (int k = 0; k < 8; k++) { o.a.v[k] = lds[k * TW + src]; o.b.v[k] = lds[(8 + k) * TW + src]; o.c.v[k] = lds[(16 + k) * TW + src]; }
            t = triple_add(t, o);
        }
        __syncthreads();
    }
    return t;
}
__global__ void __launch_bounds__(TW, ZKP_LAT_WAVES) k_poly_sum(BpView V) {
    ZKP_RAISE_PRIO();
    __shared__ uint32_t lds[24 * TW];
    const uint32_t job = blockIdx.x * 8 + (threadIdx.x & 7u), part = threadIdx.x >> 3;
    const bool active = job < V.M;
    ScTriple t{sc_zero(), sc_zero(), sc_zero()};
    if (active) t = step_poly_sum_part(V, part, job);
    t = triple_tree8(t, lds);
    if (active && part == 0) step_poly_sum_finish(V, job, t);
}
__global__ void __launch_bounds__(TB) k_lr_init(BpView V) {
    ZKP_RAISE_PRIO();
    const uint32_t job = blockIdx.x * TB + threadIdx.x;
    if (job < V.M) step_lr_init(V, blockIdx.y, job);
}
__global__ void __launch_bounds__(TB, ZKP_LAT_WAVES) k_round_prep(BpView V, uint32_t r) {
    ZKP_RAISE_PRIO();
    const uint32_t job = blockIdx.x * TB + threadIdx.x;
    if (job < V.M) step_round_prep(V, r, blockIdx.y, job);
}
__global__ void __launch_bounds__(TW, ZKP_LAT_WAVES) k_round_sum(BpView V, uint32_t r) {
    ZKP_RAISE_PRIO();      // <= 64 additions per lane: the 8-lane split measured slower here
    const uint32_t job = blockIdx.x * TW + threadIdx.x;
    if (job < V.M) step_round_sum(V, r, job);
}
// Register budget of the heavy chain kernels (transcripts, partial sums, ristretto encoding), as waves per SIMD: 1 = whatever the compiler
// likes (k_encode: 273 registers, k_sum_t: 209).  A/B knob (-DZKP_BP_CHAIN_WAVES=3 / 4: at most 168 / 128 registers, so that a wave starts
// in the space ONE retiring gather wave frees instead of waiting for two).  Measured in round 4 on the mixed batch: 12.26-12.37 ms
// unconstrained, 12.26 at 3, 12.41 at 4 (range-only batches 2 % slower with either cap): no gain, left at 1.
#ifndef ZKP_BP_CHAIN_WAVES
#define ZKP_BP_CHAIN_WAVES 1
#endif
// transcript steps: STROBE image of lane t at lds[i * TW + t] (conflict-free: consecutive lanes, consecutive banks)
__global__ void __launch_bounds__(TW, ZKP_BP_CHAIN_WAVES) k_transcript1(BpView V) {
    ZKP_RAISE_PRIO();
    __shared__ uint32_t lds[50 * TW];
    const uint32_t job = blockIdx.x * TW + threadIdx.x;
    Strobe s; s.base = lds + threadIdx.x; s.stride = TW; s.pos = 0; s.pos_begin = 0;
    if (job < V.M) step_transcript1(V, job, s);
}
__global__ void __launch_bounds__(TW, ZKP_BP_CHAIN_WAVES) k_transcript2(BpView V) {
    ZKP_RAISE_PRIO();
    __shared__ uint32_t lds[50 * TW];
    const uint32_t job = blockIdx.x * TW + threadIdx.x;
    Strobe s; s.base = lds + threadIdx.x; s.stride = TW; s.pos = 0; s.pos_begin = 0;
    if (job < V.M) step_transcript2(V, job, s);
}
__global__ void __launch_bounds__(TW, ZKP_BP_CHAIN_WAVES) k_transcript_round(BpView V, uint32_t r) {
    ZKP_RAISE_PRIO();
    __shared__ uint32_t lds[50 * TW];
    const uint32_t job = blockIdx.x * TW + threadIdx.x;
    Strobe s; s.base = lds + threadIdx.x; s.stride = TW; s.pos = 0; s.pos_begin = 0;
    if (job < V.M) step_transcript_round(V, r, job, s);
}
// Sum of a target's chunk partials: 8 lanes cooperate on one (point, proof) -- each sums every 8th partial, then a
// 3-level tree through LDS -- followed by k_encode (one lane per point, full waves) for the ristretto encoding.
// (Point addition is associative and the encoding canonical, so the bytes equal reduce_encode_thread's sequential sum,
// which the host emulation uses.)
__global__ void __launch_bounds__(TW, ZKP_BP_CHAIN_WAVES) k_encode(ReduceView R, const uint32_t* sums) {
    ZKP_RAISE_PRIO();
    const uint32_t row = blockIdx.x * TW + threadIdx.x, target = blockIdx.y;
    if (row >= R.rows) return;
    sc e; ge_ristretto_encode(e.v, ld_ge(sums, target, row, R.rows));
    st_sc(R.enc, target, row, R.rows, e);
    if (R.out_off != nullptr && target == 0) put_bytes(R.out + R.out_off[row], e.v, 8);
}

// ------------------------------------------------------------------------------------------------
// Fixed-base multiscalar multiplication, lane = proof: k_msm_gather<EdGather> (msm_kernel.h, edg_kernels.hip) over the radix-2^16
// tables of edg.h in HBM; 256-lane workgroups of independent waves, one chunk of (generator, window) steps each, workgroups that share
// a chunk on one XCD.  EdMsm names the point type for the partial-sum kernel.
struct EdMsm {      // edwards25519 affine-Niels tables, extended-coordinate accumulator (Bulletproofs path)
    static constexpr uint32_t ACC_W = GE_W;
    using Acc = ge;
    static __device__ __forceinline__ Acc identity() { return ge_identity(); }
    static __device__ __forceinline__ void store(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const Acc& a) { st_ge(p, idx, row, rows, a); }
    static __device__ __forceinline__ Acc load(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) { return ld_ge(p, idx, row, rows); }
    static __device__ __forceinline__ Acc add(const Acc& a, const Acc& b) { return ge_add(a, b); }
};

static constexpr uint32_t ED_SUM_ROWS = 32, ED_SUM_TB = 256;      // 8 slices per row; one wave per SIMD
template __global__ void k_sum_t<EdMsm, ED_SUM_ROWS, ED_SUM_TB, ZKP_BP_CHAIN_WAVES>(ReduceView, uint32_t*);

// ================================================================================================ host
namespace {

thread_local std::string t_err;
int fail(int code, const std::string& msg) { t_err = msg; return code; }
// ---- Exception barrier of the C ABI ("per-item status code + message; never abort" -- batch.rs:126-130 turns every failure
// into an Err, error_handling.rs:39-50 maps it for Python).  Every exported function is a function-try-block that ends in one of these
// handlers: a std::bad_alloc / std::system_error / anything else raised by the host logic (vectors sized from the caller's n, strings,
// worker threads, mutexes) becomes ZKP_HIP_E_RUNTIME with a message for zkp_hip_last_error instead of unwinding through a C frame into
// the caller's Rust (undefined behaviour).  RAII releases what the call held (Bind, DevScope, vectors); the library stays usable.
int fail_nothrow(int code, const char* what, const char* detail) noexcept {
    try { t_err.assign(what); if (detail) t_err.append(detail); }
    catch (...) { t_err.clear(); try { t_err.assign("out of memory"); } catch (...) {} }      // (fits the small-string buffer)
    return code;
}
#define ZKP_API_CATCH_RET(on_fail)                                                                                                  \
    catch (const std::bad_alloc&) { (void)fail_nothrow(ZKP_HIP_E_RUNTIME, "out of host memory", nullptr); on_fail; }                   \
    catch (const std::exception& e_) { (void)fail_nothrow(ZKP_HIP_E_RUNTIME, "C++ exception at the C ABI: ", e_.what()); on_fail; }    \
    catch (...) { (void)fail_nothrow(ZKP_HIP_E_RUNTIME, "unknown C++ exception at the C ABI", nullptr); on_fail; }
#define ZKP_API_CATCH_INT ZKP_API_CATCH_RET(return ZKP_HIP_E_RUNTIME)
#define ZKP_API_CATCH_ZERO ZKP_API_CATCH_RET(return 0)
#define ZKP_API_CATCH_VOID ZKP_API_CATCH_RET(return)
// the same for the body of a host worker thread (an exception that leaves a std::thread's function is std::terminate)
template <class F> int guarded(F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail_nothrow(ZKP_HIP_E_RUNTIME, "out of host memory", nullptr); }
    catch (const std::exception& e_) { return fail_nothrow(ZKP_HIP_E_RUNTIME, "C++ exception in a shard worker: ", e_.what()); }
    catch (...) { return fail_nothrow(ZKP_HIP_E_RUNTIME, "unknown C++ exception in a shard worker", nullptr); }
}
// Largest batch one call accepts: beyond it the workspaces (~110 KB of HBM per range op) cannot exist on a 288 GB device anyway, and
// host vectors / pinned staging are sized from n -- an absurd n is an argument error, not an allocation attempt.
constexpr uint64_t ZKP_MAX_BATCH_OPS = 1ull << 22;
constexpr uint64_t ZKP_MAX_LIST_VALUES = 1ull << 28;      // threshold / consistency value lists of one call, summed
int check_batch_size(uint64_t n) {
    if (n > ZKP_MAX_BATCH_OPS) return fail(ZKP_HIP_E_ARGUMENT, "batch too large: at most 4194304 operations per call (split the batch)");
    return 0;
}
// counts[] of a threshold / consistency call: the lists are read by index sums, so their total is bounded BEFORE anything is read or sized from it
int check_list_total(uint64_t n, const uint32_t* counts) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) { total += counts[i]; if (total > ZKP_MAX_LIST_VALUES) return fail(ZKP_HIP_E_ARGUMENT, "value lists too long: at most 2^28 values per call (split the batch)"); }
    return 0;
}
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(ZKP_HIP_E_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Per-call device buffers (staging of host inputs / outputs): handed back when the call returns, on every path (HIP_TRY
// returns early on errors).  Blocks are kept in a small pool instead of going through hipMalloc / hipFree on every call
// (hipFree synchronises the device; a one-proof call spent a third of its time there): a block is reused for requests
// between half its size and its size, and the pool is emptied when it holds more than DEV_POOL_LIMIT bytes of idle blocks.
struct DevPool {
    struct Block { void* p; size_t bytes; bool busy; };
    std::vector<Block> blocks;
    std::mutex mu;
    static constexpr size_t DEV_POOL_LIMIT = (size_t)1 << 30;
    const bool bypass = getenv("ZKP_HIP_NO_POOL") != nullptr;        // debugging knob: plain hipMalloc / hipFree per call
    hipError_t take(void** out, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        if (!bypass) for (auto& b : blocks) if (!b.busy && b.bytes >= bytes && b.bytes / 2 <= bytes) { b.busy = true; *out = b.p; return hipSuccess; }
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) {                       // out of memory: drop the idle blocks and try once more
            trim_locked(0);
            (void)hipGetLastError();
            e = hipMalloc(&q, bytes);
            if (e != hipSuccess) { *out = nullptr; return e; }
        }
        blocks.push_back({q, bytes, true});
        *out = q;
        return hipSuccess;
    }
    void give(void* p) {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& b : blocks) if (b.p == p) b.busy = false;
        trim_locked(bypass ? 0 : DEV_POOL_LIMIT);
    }
    void trim_locked(size_t keep) {
        size_t idle = 0; for (auto& b : blocks) if (!b.busy) idle += b.bytes;
        if (idle <= keep) return;
        std::vector<Block> rest;
        for (auto& b : blocks) { if (b.busy) rest.push_back(b); else (void)hipFree(b.p); }
        blocks.swap(rest);
    }
    void release_all() { std::lock_guard<std::mutex> lk(mu); trim_locked(0); }
};
DevPool& dev_pool();            // the pool of the shard the calling thread is bound to
void dev_scope_quiesce();       // waits for that shard's own stream: a block must be idle before it goes back to the pool
struct DevScope {
    std::vector<void*> owned;
    DevScope() = default;
    DevScope(const DevScope&) = delete;
    DevScope& operator=(const DevScope&) = delete;
    ~DevScope() { release(); }
    void release() { if (!owned.empty()) dev_scope_quiesce(); for (void* q : owned) dev_pool().give(q); owned.clear(); }
    template <class T> hipError_t alloc(T** out, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = dev_pool().take(&q, bytes ? bytes : 1);
        if (e == hipSuccess) owned.push_back(q);
        *out = static_cast<T*>(q);
        return e;
    }
};

struct DevLayout {
    uint16_t *slot_base = nullptr, *chunk_begin = nullptr, *chunk_win0 = nullptr, *chunk_nwin = nullptr, *target_chunk_begin = nullptr;
    uint8_t* slot_nwin = nullptr;
    uint32_t nslots = 0, nchunks = 0, ntargets = 0, max_chunk_windows = 0, max_target_chunks = 0;
    uint64_t adds_per_row = 0;   // sum of nwin = point additions per proof in this launch
    uint32_t *steps = nullptr, *chunk_step0 = nullptr;      // k_msm_gather's flat step list (make_gather_steps); only for layouts that kernel walks
};
// candidate chunkings of one launch type: slot-aligned chunks of 32*T windows (T = 1..8) and window-granular "even"
// chunkings with a given chunk count; the launch picks the one whose grid best fills the resident workgroup slots
constexpr int MAXT = 8;
struct LayoutSet { std::vector<DevLayout> cand; uint32_t max_chunks = 0; };

struct SubBatch {
    hipStream_t stream = nullptr;
    hipEvent_t start = nullptr, done = nullptr;
    hipStream_t side = nullptr;                    // the commitment tasks of a batch (independent of its proofs) run beside the first phase
    hipEvent_t side_go = nullptr, side_done = nullptr;
    bool used = false;                             // `done` has been recorded at least once
    bool borrowed = false;                         // stream / side belong to slot 0 (the second slot is a second WORKSPACE on the same streams)
    void* ws = nullptr;
    uint32_t capM = 0, capC = 0, cap_chunks = 0;
};
constexpr uint32_t NSLOTS = 2;      // device-pointer calls alternate between two sets of streams + workspace, so a caller that
                                    // feeds batches from two of its own streams keeps two batches in flight (bench.py --pipeline 2)

// One shard = one HIP device context of the library: tables, workspaces, streams, loaded keys.  The registry maps shard
// numbers to HIP devices (the same HIP device may back two shards: two independent contexts, which is how the one-GPU
// tests exercise the multi-GPU path).  Each ABI call binds the calling thread to one shard (`Bind`) and holds that shard's
// mutex; calls on different shards run concurrently, and zkp_hip_process_batch / the staged-batch entry points drive
// every shard from one host worker thread each (batch_impl.inc).
struct G16State; struct StarkState; struct VfyState; struct BatchState;
struct Device {
    int index = 0, hip_dev = 0;
    std::mutex mu;
    bool ready = false;
    uint64_t generation = 0;            // bumped by zkp_hip_shutdown: staged batches of an earlier life own nothing any more
    int num_cu = 256;
    int msm_prio_now = 0;               // set by the mixed-batch scheduler while it enqueues the Bulletproofs chain of a batch that also holds Groth16 work: the ed25519 MSM waves raise their issue priority
    int cus_now = 0;                    // CUs the launches being enqueued may use (0 = all): set by the mixed-batch scheduler while it enqueues a variant on CU-masked streams
    std::vector<SubBatch> subm;         // [lane]: Bulletproofs streams + workspace confined to the Bulletproofs CU partition of a mixed batch
    hipStream_t stream = nullptr;
    uint32_t* d_edg_table = nullptr;    // radix-2^16 tables, gathered per lane from HBM (edg.h: every MSM of the prover)
    int edg_blocks_per_cu = 3;
    // MSM chunkings: phase 1 and the inner-product rounds depend on the proofs' bit width n = 8 << w (w = 0..3); the
    // 64-bit family is built at init, narrower ones on first use
    struct Family { LayoutSet p1, rd[6]; bool ready = false; };
    Family fam[4];
    LayoutSet p2, ct;
    uint32_t max_chunks = 0;
    std::vector<SubBatch> sub;          // [slot * nsub + h]
    uint32_t nsub = 1, next_slot = 0;
    DevPool pool;
    // the other parts of the library keep their per-shard state behind these (created on first use, freed by shutdown)
    G16State* g16 = nullptr; StarkState* stark = nullptr; VfyState* vfy = nullptr; BatchState* batch = nullptr;
    // profiling (bench.py's roofline): every launch of the three MSM kernels bracketed by events on its own stream
    bool profiling = false;
    struct KProf {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used = 0;
        double ms = 0; uint64_t launches = 0, adds = 0;
    };
    KProf prof[3];                      // ZKP_HIP_KERNEL_MSM_ED25519 / _BN254_G1 / _BN254_G2
    struct Trace* trace = nullptr;      // ZKP_HIP_TRACE=<file>: a timeline of every launch of a mixed batch (tools/trace_timeline.py)
    struct ShardWorker* worker = nullptr;      // the host thread that drives this shard in multi-shard calls (created on first use, parked between calls)
};
// One parked host thread per shard: a multi-shard batch call (stage, prove, fetch = three fan-outs per batch) hands each shard's share to
// that shard's worker instead of creating and joining a std::thread per shard per fan-out.  The thread touches HIP only inside a job; it is
// never joined (nothing of this library is destroyed from exit(), see the registry below) and survives zkp_hip_shutdown parked.
struct ShardWorker {
    // A FIFO of jobs, each with its own completion flag: callers on different host threads (one staging batch N + 1 while another waits
    // for batch N) may post to the same shard's worker at once; every caller waits for ITS tickets only, so no job is overwritten and
    // nobody returns on somebody else's completion (round 3 kept ONE job slot behind a lock that every call site had its own copy of).
    struct Job { std::function<void()> fn; bool done = false; };
    using Ticket = std::shared_ptr<Job>;
    std::mutex mu; std::condition_variable cv;
    std::deque<Ticket> queue;
    std::thread th;
    ShardWorker() : th([this]() { run(); }) { th.detach(); }
    void run() {
        for (;;) {
            Ticket t;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this]() { return !queue.empty(); }); t = queue.front(); queue.pop_front(); }
            t->fn();                                   // (posted bodies never throw: for_each_shard wraps them in guarded())
            { std::lock_guard<std::mutex> lk(mu); t->done = true; }
            cv.notify_all();
        }
    }
    Ticket post(std::function<void()> f) {
        Ticket t = std::make_shared<Job>(); t->fn = std::move(f);
        { std::lock_guard<std::mutex> lk(mu); queue.push_back(t); }
        cv.notify_all();
        return t;
    }
    void wait(const Ticket& t) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return t->done; }); }
};
std::mutex g_worker_create_mu;          // one lock for every call site (workers are created once per shard)
// event pair around one profiled launch (nullptrs when profiling is off)
int prof_begin(Device::KProf& K, hipStream_t st, hipEvent_t* e1);
void prof_end(Device::KProf& K, hipStream_t st, hipEvent_t e1, uint64_t adds);
// Everything below is allocated once and never destroyed: no destructor of this library runs from exit() (the HIP
// runtime and profiler tools tear themselves down there in an order we do not control); device resources are released
// by zkp_hip_shutdown, which an atexit hook registered at the first initialisation calls while the runtime is alive.
struct Registry {
    std::mutex mu;
    std::vector<Device*> shards;
    bool hooked = false;
};
Registry& registry() { static Registry* r = new Registry(); return *r; }
thread_local Device* t_dev = nullptr;      // shard the calling thread is bound to for the duration of an ABI call
thread_local int t_sel = 0;                // shard the per-variant entry points of this thread use (zkp_hip_use_device)
Device& dev() { return *t_dev; }
DevPool& dev_pool() { return dev().pool; }
void dev_scope_quiesce() { if (dev().stream) (void)hipStreamSynchronize(dev().stream); }
uint32_t g_budget_request = 0;     // 0 = choose per launch
double g_fill = 1.0;               // benchmarking knob: scales the resident-workgroup count the Bulletproofs MSM chunking aims at
uint32_t g_subbatches = 1;         // >1: independent slices on separate streams (measured slower on MI355X: see DESIGN.md)
// Stream priorities (mixed batches run their variants on separate streams).  0 = greatest priority of the device, 1 = default,
// 2 = least.  What counts is the order: the Bulletproofs pipeline -- a chain of ~50 dependent launches, most of them short -- above
// the Groth16 streams with their long MSM grids (4096-op mixed batch, same box: 13.8-13.9 ms with Bulletproofs above Groth16, 16.0 ms
// with everything at one level).  Bulletproofs and the STARK sit at the DEFAULT level and Groth16 at the least: with the Bulletproofs
// streams at the greatest level (rounds 1-2) a range-only batch entered through the host-buffer API, whose copies and fork / join run
// on a default-level stream, lost ~50 us between consecutive kernels once a mixed batch had created the other levels' queues
// (5.9-6.2 against 4.2 ms per 1024 ops; launch traces in round 3) -- the "C2 regression" of round 2's bench line was that, hit on
// every second call because the calls alternated between two stream sets.
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
int stream_priority(int level) {
    int least = 0, greatest = 0; (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    return level == 0 ? greatest : level == 2 ? least : (least + greatest) / 2;
}
int bp_priority_level() { static const int v = env_int("ZKP_HIP_BP_PRIORITY", 1); return v; }
int stark_priority_level() { static const int v = env_int("ZKP_HIP_STARK_PRIORITY", 1); return v; }
int g16_priority_level() { static const int v = env_int("ZKP_HIP_G16_PRIORITY", 2); return v; }

int prof_begin(Device::KProf& K, hipStream_t st, hipEvent_t* e1) {
    *e1 = nullptr;
    if (!dev().profiling) return 0;
    if (K.ev_used == K.ev_pool.size()) { hipEvent_t a, b; HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b)); K.ev_pool.push_back({a, b}); }
    HIP_TRY(hipEventRecord(K.ev_pool[K.ev_used].first, st));
    *e1 = K.ev_pool[K.ev_used].second; K.ev_used++;
    return 0;
}
void prof_end(Device::KProf& K, hipStream_t st, hipEvent_t e1, uint64_t adds) {
    if (!e1) return;
    (void)hipEventRecord(e1, st); K.launches++; K.adds += adds;
}

// ---- launch timeline (debugging / profiling aid, off unless ZKP_HIP_TRACE names a file).  rocprofv3's kernel trace costs ~30 us of
// host time per dispatch, which delays the later chains of a mixed batch by milliseconds and so changes the very schedule one wants to
// see; two event records around a launch cost ~2 us.  A record holds "the stream reached this launch" and "the kernel finished".
struct Trace {
    struct Rec { const char* name; hipStream_t st; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool; size_t used = 0;
    hipEvent_t origin = nullptr; bool have_origin = false;
    std::string path;
    hipEvent_t take() { if (used == pool.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); pool.push_back(e); } return pool[used++]; }
};
Trace* trace_state() {
    Device& D = dev();
    if (D.trace) return D.trace->path.empty() ? nullptr : D.trace;
    D.trace = new Trace();
    const char* v = getenv("ZKP_HIP_TRACE");
    if (v && *v) D.trace->path = v;
    return D.trace->path.empty() ? nullptr : D.trace;
}
void trace_origin(hipStream_t st) {
    Trace* T = trace_state(); if (!T) return;
    if (!T->origin) (void)hipEventCreate(&T->origin);
    (void)hipEventRecord(T->origin, st); T->have_origin = true;
}
struct TraceMark {
    Trace* T; Trace::Rec r;
    TraceMark(const char* name, hipStream_t st) : T(trace_state()) { if (T) { r = {name, st, T->take(), T->take()}; (void)hipEventRecord(r.a, st); } }
    ~TraceMark() { if (T) { (void)hipEventRecord(r.b, r.st); T->recs.push_back(r); } }
};
#define ZKP_TRACED(name, st, ...) do { TraceMark tm_(name, st); __VA_ARGS__; } while (0)
// appends one JSON line per batch: [[name, stream, t_reached_ms, t_done_ms], ...] relative to the batch's first enqueue
void trace_dump() {
    Trace* T = trace_state(); if (!T || T->recs.empty()) return;
    (void)hipDeviceSynchronize();
    FILE* f = fopen(T->path.c_str(), "a");
    if (f) {
        std::map<hipStream_t, int> ids;
        fputs("[", f);
        bool first = true;
        for (auto& r : T->recs) {
            float ta = 0, tb = 0;
            if (!T->have_origin || hipEventElapsedTime(&ta, T->origin, r.a) != hipSuccess || hipEventElapsedTime(&tb, T->origin, r.b) != hipSuccess) continue;
            const int id = ids.emplace(r.st, (int)ids.size()).first->second;
            fprintf(f, "%s[\"%s\", %d, %.4f, %.4f]", first ? "" : ", ", r.name, id, ta, tb); first = false;
        }
        fputs("]\n", f);
        fclose(f);
    }
    (void)hipGetLastError();
    T->recs.clear(); T->used = 0; T->have_origin = false;
}
void trace_release() {
    Trace* T = dev().trace; if (!T) return;
    for (auto e : T->pool) (void)hipEventDestroy(e);
    if (T->origin) (void)hipEventDestroy(T->origin);
    delete T; dev().trace = nullptr;
}

int upload_layout(DevLayout& D, const MsmLayout& L, const GatherShape* shape = nullptr, const uint16_t* slot_scalar = nullptr) {
    D.nslots = L.nslots(); D.nchunks = L.nchunks(); D.ntargets = L.ntargets();
    D.adds_per_row = 0; for (uint8_t x : L.slot_nwin) D.adds_per_row += x;
    D.max_chunk_windows = 0;
    for (uint32_t c = 0; c < D.nchunks; c++) if (L.chunk_nwin[c] > D.max_chunk_windows) D.max_chunk_windows = L.chunk_nwin[c];
    D.max_target_chunks = 0;
    for (uint32_t t = 0; t < D.ntargets; t++) {
        const uint32_t k = L.target_chunk_begin[t + 1] - L.target_chunk_begin[t];
        if (k > D.max_target_chunks) D.max_target_chunks = k;
    }
    HIP_TRY(hipMalloc(&D.slot_base, L.slot_base.size() * 2));
    HIP_TRY(hipMalloc(&D.chunk_begin, L.chunk_begin.size() * 2 + 2));
    HIP_TRY(hipMalloc(&D.chunk_win0, L.chunk_win0.size() * 2 + 2));
    HIP_TRY(hipMalloc(&D.chunk_nwin, L.chunk_nwin.size() * 2 + 2));
    HIP_TRY(hipMalloc(&D.target_chunk_begin, L.target_chunk_begin.size() * 2));
    HIP_TRY(hipMalloc(&D.slot_nwin, L.slot_nwin.size()));
    HIP_TRY(hipMemcpy(D.slot_base, L.slot_base.data(), L.slot_base.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.chunk_begin, L.chunk_begin.data(), L.chunk_begin.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.chunk_win0, L.chunk_win0.data(), L.chunk_win0.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.chunk_nwin, L.chunk_nwin.data(), L.chunk_nwin.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.target_chunk_begin, L.target_chunk_begin.data(), L.target_chunk_begin.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(D.slot_nwin, L.slot_nwin.data(), L.slot_nwin.size(), hipMemcpyHostToDevice));
    if (shape) {
        std::vector<uint32_t> steps, step0;
        if (!make_gather_steps(L, slot_scalar, *shape, steps, step0)) return fail(ZKP_HIP_E_UNSUPPORTED, "window tables of more than 2^32 entries");
        HIP_TRY(hipMalloc(&D.steps, steps.size() * 4)); HIP_TRY(hipMalloc(&D.chunk_step0, step0.size() * 4));
        HIP_TRY(hipMemcpy(D.steps, steps.data(), steps.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(D.chunk_step0, step0.data(), step0.size() * 4, hipMemcpyHostToDevice));
    }
    return 0;
}
void free_layout(DevLayout& D) {
    (void)hipFree(D.slot_base); (void)hipFree(D.chunk_begin); (void)hipFree(D.chunk_win0); (void)hipFree(D.chunk_nwin); (void)hipFree(D.target_chunk_begin); (void)hipFree(D.slot_nwin);
    (void)hipFree(D.steps); (void)hipFree(D.chunk_step0);
    D = DevLayout();
}
// chunkings of one launch over the HBM tables of edg.h (every ed25519 MSM of the library -- prover and, since round 4, verifier -- reads them)
int upload_set(LayoutSet& S, const std::vector<SlotList>& targets) {
    uint32_t total = 0; for (auto& t : targets) for (auto& sl : t) total += sl.second;
    auto push = [&](const MsmLayout& L) -> int {
        S.cand.emplace_back();
        const GatherShape edg_shape{EDG_NENT, EDG_NWIN * EDG_NENT, 0u, DIGW};      // (digit rows keep their 13-word pitch)
        int rc = upload_layout(S.cand.back(), L, &edg_shape); if (rc) return rc;
        if (S.cand.back().nchunks > S.max_chunks) S.max_chunks = S.cand.back().nchunks;
        return 0;
    };
    for (int T = 1; T <= MAXT; T++) { int rc = push(make_layout(targets, 32u * T)); if (rc) return rc; }
    static const uint32_t counts[] = {3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40, 42, 48, 51, 56, 64, 72, 80, 85, 96, 112, 128, 144, 160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 1024};
    for (uint32_t c : counts) { if (c > total || c < targets.size()) continue; int rc = push(make_layout_even(targets, c)); if (rc) return rc; }
    return 0;
}
void free_set(LayoutSet& S) { for (auto& d : S.cand) free_layout(d); S.cand.clear(); S.max_chunks = 0; }

// Chunk size for one launch: the grid is nchunks * ceil(rows/256) workgroups, edg_blocks_per_cu * num_cu of which are
// resident at a time; cost = (#rounds of resident workgroups) * (windows per workgroup) + the serial partial-sum tail.
const DevLayout& pick_layout(const LayoutSet& S, uint32_t rows) {
    if (g_budget_request >= 10000) {             // benchmarking knob: the even layout whose chunk count is closest to (request - 10000)
        const uint32_t want = g_budget_request - 10000; size_t best = MAXT; uint32_t bd = 0xffffffffu;
        for (size_t i = MAXT; i < S.cand.size(); i++) { const uint32_t d = S.cand[i].nchunks > want ? S.cand[i].nchunks - want : want - S.cand[i].nchunks; if (d < bd) { bd = d; best = i; } }
        return S.cand[best < S.cand.size() ? best : 0];
    }
    if (g_budget_request) { int T = (int)(g_budget_request / 32); if (T < 1) T = 1; if (T > MAXT) T = MAXT; return S.cand[T - 1]; }
    // Take the window-granular layout that minimises rounds x (windows per workgroup + per-workgroup overhead) + the partial-sum
    // work that grows with the chunk count; g_fill scales the resident count (benchmarking knob, default 1).
    static const int fill_pct = env_int("ZKP_HIP_BP_FILL", 100);      // tuning knob: size the MSM grids for this percentage of the CUs
    // The gather launches run 256-lane workgroups of four independent waves, edg_blocks_per_cu of them per CU:
    // no strict rounds, but the same trade -- more chunks fill the chip and shorten a lane's chain of additions, and every chunk is one
    // more partial point per proof for k_sum_t (a 9-product addition against the 7 of a table step).
    const uint32_t tb = edg_msm_rows_per_block();
    const double per_cu = (double)dev().edg_blocks_per_cu;
    const double resident = g_fill * (fill_pct / 100.0) * (double)(dev().cus_now ? dev().cus_now : dev().num_cu) * per_cu;
    const uint32_t groups = (rows + tb - 1) / tb;
    size_t best = MAXT; double best_cost = 1e300;
    for (size_t i = MAXT; i < S.cand.size(); i++) {
        const uint32_t nc = S.cand[i].nchunks;
        const double rounds = std::ceil((double)nc * groups / resident);
        const double cost = rounds * ((double)S.cand[i].max_chunk_windows + 1.0) + 1.3 * S.cand[i].max_target_chunks / 8.0;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return S.cand[best < S.cand.size() ? best : 0];
}

// chunkings of the launches whose generator set depends on the bit width (lg = log2 n, 3..6)
int ensure_family(uint32_t lg) {
    Device::Family& F = dev().fam[lg - 3];
    if (F.ready) return 0;
    int rc;
    if ((rc = upload_set(F.p1, targets_phase1(1u << lg, EDG_NWIN, EDG_NWIN_U64)))) return rc;
    for (uint32_t r = 0; r < lg; r++) if ((rc = upload_set(F.rd[r], targets_round(r, 1u << lg, EDG_NWIN)))) return rc;
    if (F.p1.max_chunks > dev().max_chunks) dev().max_chunks = F.p1.max_chunks;
    for (uint32_t r = 0; r < lg; r++) if (F.rd[r].max_chunks > dev().max_chunks) dev().max_chunks = F.rd[r].max_chunks;
    F.ready = true;
    return 0;
}
bool bits_to_lg(uint32_t n_bits, uint32_t* lg) {     // RangeProof::prove_single accepts 8, 16, 32, 64 (InvalidBitsize otherwise)
    for (uint32_t k = 3; k <= 6; k++) if (n_bits == (1u << k)) { *lg = k; return true; }
    return false;
}

// The prover's tables (edg.h): 130 generators x 16 windows x 32 768 affine-Niels entries in 128-byte slots = 8.7 GB of HBM, computed on the
// device from the 130 generators (the host derives only those: RFC 9496 one-way map, bp_layout.h) and checked slot against slot before the
// shard is declared ready.  Takes ~0.1 s.
// One table per physical GPU in the process: shards registered on the same HIP device share it (like the Groth16 key tables).
struct EdgTables { std::mutex mu; struct E { int hip_dev; uint32_t* table; int refs; }; std::vector<E> v; };
EdgTables& edg_tables() { static EdgTables* r = new EdgTables(); return *r; }
void release_edg_table() {
    Device& D = dev();
    if (!D.d_edg_table) return;
    EdgTables& R = edg_tables();
    std::lock_guard<std::mutex> lk(R.mu);
    for (auto& e : R.v) if (e.table == D.d_edg_table && e.refs > 0 && --e.refs == 0) { (void)hipFree(e.table); e.table = nullptr; }
    D.d_edg_table = nullptr;
}
int build_edg_table() {
    Device& D = dev();
    if (D.d_edg_table) return 0;
    EdgTables& R = edg_tables();
    std::lock_guard<std::mutex> registry_lock(R.mu);          // held across the build: a second shard of this GPU waits and then shares
    {
        const uint32_t occ = edg_msm_blocks_per_cu();
        D.edg_blocks_per_cu = occ ? (int)occ : 3;
    }
    for (auto& e : R.v) if (e.hip_dev == D.hip_dev && e.refs > 0) { e.refs++; D.d_edg_table = e.table; return 0; }
    ge gens[NBASE]; host_generators(gens);
    std::vector<uint32_t> gw((size_t)NBASE * GE_W);
    for (uint32_t b = 0; b < NBASE; b++) st_ge(gw.data(), b, 0, 1, gens[b]);
    uint32_t *d_gens = nullptr, *d_scratch = nullptr; int* d_bad = nullptr;
    const size_t table_bytes = edg_table_words() * 4, scratch_bytes = edg_build_scratch_words() * 4;
    if (hipMalloc(&D.d_edg_table, table_bytes) != hipSuccess) {
        (void)hipGetLastError(); D.d_edg_table = nullptr;
        char msg[200]; snprintf(msg, sizeof msg, "out of device memory allocating the %llu MB of generator window tables", (unsigned long long)(table_bytes >> 20));
        return fail(ZKP_HIP_E_RUNTIME, msg);
    }
    auto drop = [&]() { (void)hipFree(d_gens); (void)hipFree(d_scratch); (void)hipFree(d_bad); };
    auto give_up = [&](int code, const std::string& m) { drop(); (void)hipFree(D.d_edg_table); D.d_edg_table = nullptr; return fail(code, m); };
    hipError_t e = hipMalloc(&d_gens, gw.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&d_scratch, scratch_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(d_gens, gw.data(), gw.size() * 4, hipMemcpyHostToDevice, D.stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), D.stream);
    if (e != hipSuccess) return give_up(ZKP_HIP_E_RUNTIME, std::string("generator tables: ") + hipGetErrorString(e));
    edg_launch_build(d_gens, D.d_edg_table, d_scratch, d_bad, D.stream);
    int bad = -1;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, D.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
    if (e != hipSuccess) return give_up(ZKP_HIP_E_RUNTIME, std::string("generator tables: ") + hipGetErrorString(e));
    if (bad != 0) { char msg[160]; snprintf(msg, sizeof msg, "generator window tables failed their self-check (%d inconsistent slots)", bad); return give_up(ZKP_HIP_E_RUNTIME, msg); }
    drop();
    R.v.push_back({D.hip_dev, D.d_edg_table, 1});
    return 0;
}

// brings the bound shard up (caller holds its mutex and has made its HIP device current)
int init_device() {
    Device& D = dev();
    if (D.ready) return 0;
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, D.hip_dev));
    D.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
    int rc;
    if ((rc = build_edg_table())) return rc;
    D.max_chunks = 0;
    if ((rc = upload_set(D.p2, targets_phase2(EDG_NWIN)))) return rc;
    if ((rc = upload_set(D.ct, targets_ctask(EDG_NWIN, EDG_NWIN_U64)))) return rc;
    D.max_chunks = D.p2.max_chunks;
    if ((rc = ensure_family(6))) return rc;
    uint32_t ns = g_subbatches; if (ns <= 1) ns = (uint32_t)env_int("ZKP_HIP_BP_SUBBATCHES", 1); if (ns < 1) ns = 1; if (ns > 8) ns = 8;
    D.nsub = ns;
    D.sub.resize((size_t)NSLOTS * ns);       // streams and events of a slot exist from its first use (ensure_sub)
    D.ready = true;
    return 0;
}
// ---- CU partition of a mixed batch (experiment, OFF by default: ZKP_HIP_BP_CUS = CUs per XCD for the Bulletproofs streams).  The
// Bulletproofs prover is a chain of ~50 dependent launches whose MSM workgroups need whole CUs (1024 lanes, 120 KB of LDS) and whose
// lane = proof kernels need 120-256 VGPRs; next to the Groth16 gather kernels (three 136-VGPR waves per SIMD, workgroups that live
// 1.5-2.5 ms) each of those launches waits for workgroup slots (launch trace, round 3: first inner-product round reached at ~10 ms of a
// 14 ms step, the last rounds then run on an idle GPU).  Giving each side its own CUs (hipExtStreamCreateWithCUMask; mask bit b = CU
// b / 8 of XCD b mod 8, tools/cumask_probe.hip; every grid sized for its partition) removes the waiting but not the arithmetic: the
// Bulletproofs MSMs are 2.2 ms of whole-GPU work, so on a quarter of the CUs the chain takes 12+ ms and the step 16.3-16.8 ms (8 or 12
// CUs per XCD) against 13.7 ms unpartitioned on the same box; 6 and 10 CUs per XCD measured 21-27 ms.  Kept as a knob.
int bp_cus_per_xcd() { static const int v = env_int("ZKP_HIP_BP_CUS", 0); return v < 0 ? 0 : v > 24 ? 24 : v; }
int make_masked_stream(hipStream_t* out, bool bp_part) {
    const int per_xcd = dev().num_cu / 8, nb = bp_cus_per_xcd();
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < per_xcd * 8 && b < 256; b++) if ((b < nb * 8) == bp_part) mask[b >> 5] |= 1u << (b & 31);
    HIP_TRY(hipExtStreamCreateWithCUMask(out, 8, mask));
    return 0;
}
int partition_cus(bool bp_part) { const int nb = bp_cus_per_xcd() * 8; return bp_part ? nb : dev().num_cu - nb; }
// `share`: take the streams of that slot instead of creating a pair.  The second slot of a shard (the second of two batches in flight)
// is a second workspace on the SAME streams: its chain queues behind the first batch's chain of the same variant, which is the overlap
// one wants -- the tail of one batch under the head of the next -- without a second set of hardware queues (with its own streams the
// second lane shared queues with the first one's, and two batches in flight measured slower than one: 15.1 against 13.6 ms per batch).
int ensure_sub(SubBatch& sb, bool masked = false, SubBatch* share = nullptr) {
    if (sb.stream) return 0;
    // Built in locals and committed to `sb` only when every object exists: a failure half-way (these are created lazily, in the middle of
    // a batch enqueue) must not leave a slot that looks ready but holds null events.
    hipStream_t stream = nullptr, side = nullptr; bool borrowed = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    auto undo = [&]() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (!borrowed) { if (stream) (void)hipStreamDestroy(stream); if (side) (void)hipStreamDestroy(side); }
    };
    int rc = 0;
    if (share && share != &sb) {
        if ((rc = ensure_sub(*share, masked))) return rc;
        stream = share->stream; side = share->side; borrowed = true;
    } else if (masked) {
        if ((rc = make_masked_stream(&stream, true)) || (rc = make_masked_stream(&side, true))) { undo(); return rc; }
    } else {
        hipError_t e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, stream_priority(bp_priority_level()));
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, stream_priority(bp_priority_level()));
        if (e != hipSuccess) { undo(); return fail(ZKP_HIP_E_RUNTIME, std::string("hipStreamCreateWithPriority: ") + hipGetErrorString(e)); }
    }
    for (auto& e : ev) {
        const hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (err != hipSuccess) { e = nullptr; undo(); return fail(ZKP_HIP_E_RUNTIME, std::string("hipEventCreateWithFlags: ") + hipGetErrorString(err)); }
    }
    sb.start = ev[0]; sb.done = ev[1]; sb.side_go = ev[2]; sb.side_done = ev[3];
    sb.side = side; sb.borrowed = borrowed; sb.stream = stream;
    return 0;
}

extern "C" void zkp_hip_shutdown(void);
void batch_release_all();
void stark_release_all();
// shard `index` of the registry; registers shard 0 on HIP device 0 when nothing has been registered yet
int find_shard(int index, Device** out) {
    Registry& R = registry();
    std::lock_guard<std::mutex> lk(R.mu);
    if (R.shards.empty()) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(ZKP_HIP_E_RUNTIME, "zkp_hip_init: no HIP device available (this library has no CPU fallback)");
        Device* d = new Device(); d->index = 0; d->hip_dev = 0; R.shards.push_back(d);
    }
    if (!R.hooked) { R.hooked = true; (void)atexit(zkp_hip_shutdown); }
    if (index < 0 || index >= (int)R.shards.size()) return fail(ZKP_HIP_E_ARGUMENT, "no such device shard (zkp_hip_init / zkp_hip_init_devices register them)");
    *out = R.shards[index];
    return 0;
}
// Binds the calling thread to one shard for the duration of an ABI call: takes the shard's mutex, makes its HIP device
// current and initialises it on first use.  Calls nest on one thread only through `Bind::adopt` (worker threads of a
// multi-shard batch bind their own shard).
struct Bind {
    std::unique_lock<std::mutex> lk;
    Device* prev = nullptr; bool bound = false;
    Bind() = default;
    Bind(const Bind&) = delete;
    Bind& operator=(const Bind&) = delete;
    int open(int shard = -1) {
        Device* d = nullptr;
        int rc = find_shard(shard < 0 ? t_sel : shard, &d);
        if (rc) return rc;
        return open(d);
    }
    int open(Device* d, bool init = true) {
        lk = std::unique_lock<std::mutex>(d->mu);
        prev = t_dev; t_dev = d; bound = true;
        HIP_TRY(hipSetDevice(d->hip_dev));
        return init ? init_device() : 0;
    }
    ~Bind() { if (bound) t_dev = prev; }
};

// workspace carving ---------------------------------------------------------------------------------
struct Ws {
    JobBuf J; BpView V; CtView T;
    uint32_t *partial, *ct_partial, *ct_enc, *sums, *ct_sums;
    uint64_t* ct_off;
    int* flag;
};
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// lays the workspace out for M jobs / C commitment tasks starting at base (nullptr = size query)
size_t carve(uint8_t* base, uint32_t M, uint32_t C, uint32_t max_chunks, Ws* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { uint8_t* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t W = (size_t)8 * 4 * M;   // one scalar slot for all jobs
    const size_t WD = (size_t)DIGW * 4 * M;   // one digit slot (signed radix-1024 digits of one scalar) for all jobs
    Ws t{};
    t.J.v = (uint64_t*)take(8ull * M); t.J.seed_ix = (uint32_t*)take(4ull * M); t.J.proof_ix = (uint32_t*)take(4ull * M);
    t.J.bl_plus = (int32_t*)take(4ull * M); t.J.bl_minus = (int32_t*)take(4ull * M); t.J.kind = (uint8_t*)take(M);
    t.J.proof_off = (uint64_t*)take(8ull * M); t.J.commit_off = (uint64_t*)take(8ull * M);
    t.J.ct_v = (uint64_t*)take(8ull * C); t.J.ct_seed_ix = (uint32_t*)take(4ull * C); t.J.ct_bl_ix = (uint32_t*)take(4ull * C); t.J.ct_off = (uint64_t*)take(8ull * C);
    t.V.dig16 = 1; t.T.dig16 = 1;             // the device prover's MSMs walk the radix-2^16 tables (edg.h)
    t.V.M = M; t.V.v = t.J.v; t.V.seed_ix = t.J.seed_ix; t.V.proof_ix = t.J.proof_ix; t.V.bl_plus = t.J.bl_plus; t.V.bl_minus = t.J.bl_minus;
    t.V.kind = t.J.kind; t.V.proof_off = t.J.proof_off; t.V.commit_off = t.J.commit_off;
    t.V.tape = (uint32_t*)take(W * TAPE_SLOTS); t.V.gamma = (uint32_t*)take(W);
    t.V.d1 = (uint32_t*)take(WD * P1_NSLOTS); t.V.d2 = (uint32_t*)take(WD * P2_NSLOTS); t.V.dr = (uint32_t*)take(WD * PR_NSLOTS);
    t.V.yinvpow = (uint32_t*)take(W * 64); t.V.ypq = (uint32_t*)take(W * 32); t.V.r0 = (uint32_t*)take(W * 64); t.V.r1 = (uint32_t*)take(W * 64);
    t.V.pp = (uint32_t*)take(W * 192); t.V.ab = (uint32_t*)take(W * 256); t.V.gh = (uint32_t*)take(W * 128);
    t.V.scal = (uint32_t*)take(W * SC_NUM); t.V.tstate = (uint32_t*)take(4ull * 52 * M); t.V.enc = (uint32_t*)take(W * 3);
    t.partial = (uint32_t*)take((size_t)max_chunks * GE_W * 4 * M);
    t.sums = (uint32_t*)take((size_t)3 * GE_W * 4 * M);
    t.T.C = C; t.T.v = t.J.ct_v; t.T.seed_ix = t.J.ct_seed_ix; t.T.bl_ix = t.J.ct_bl_ix;
    t.T.digits = (uint32_t*)take((size_t)2 * DIGW * 4 * C);
    t.ct_partial = (uint32_t*)take((size_t)(dev().ct.max_chunks ? dev().ct.max_chunks : 2) * GE_W * 4 * C);
    t.ct_enc = (uint32_t*)take((size_t)8 * 4 * C);
    t.ct_sums = (uint32_t*)take((size_t)GE_W * 4 * C);
    t.ct_off = t.J.ct_off;
    t.flag = (int*)take(256);
    if (w) *w = t;
    return off;
}

int ensure_workspace(SubBatch& sb, uint32_t M, uint32_t C) {
    if (M <= sb.capM && C <= sb.capC && dev().max_chunks <= sb.cap_chunks && sb.ws) return 0;
    if (sb.ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(sb.ws)); sb.ws = nullptr; }
    const uint32_t nm = M > sb.capM ? M : sb.capM, nc = C > sb.capC ? C : sb.capC;
    const size_t bytes = carve(nullptr, nm, nc, dev().max_chunks, nullptr);
    HIP_TRY(hipMalloc(&sb.ws, bytes));
    sb.capM = nm; sb.capC = nc; sb.cap_chunks = dev().max_chunks;
    return 0;
}

int launch_msm(const DevLayout& D, uint32_t rows, const uint32_t* digits, uint32_t* partial, hipStream_t st) {
    MsmView m; m.rows = rows; m.nslots = D.nslots; m.nchunks = D.nchunks; m.table = dev().d_edg_table; m.digits = digits;
    m.slot_base = D.slot_base; m.slot_scalar = nullptr; m.slot_nwin = D.slot_nwin; m.chunk_begin = D.chunk_begin; m.chunk_win0 = D.chunk_win0; m.chunk_nwin = D.chunk_nwin; m.partial = partial; m.acc_init = nullptr;
    hipEvent_t e1 = nullptr;
    int rc = prof_begin(dev().prof[0], st, &e1);
    if (rc) return rc;
    {
        m.nwin = EDG_NWIN; m.nent = EDG_NENT; m.digw = DIGW; m.slot_ent = EDG_NWIN * EDG_NENT; m.uneven = 0;
        m.steps = D.steps; m.chunk_step0 = D.chunk_step0;
        const uint32_t tb = edg_msm_rows_per_block(), ngroups = (rows + tb - 1) / tb, nblocks = D.nchunks * ngroups;
        ZKP_TRACED("k_msm_gather<EdGather>", st, edg_launch_msm(m, ngroups, nblocks, st, dev().msm_prio_now != 0));
    }
    prof_end(dev().prof[0], st, e1, D.adds_per_row * rows);
    return 0;
}
void launch_sum_ed(const ReduceView& R, uint32_t* sums, hipStream_t st) {
    k_sum_t<EdMsm, ED_SUM_ROWS, ED_SUM_TB, ZKP_BP_CHAIN_WAVES><<<dim3((R.rows + ED_SUM_ROWS - 1) / ED_SUM_ROWS, R.ntargets), ED_SUM_TB, 0, st>>>(R, sums);
}
int launch_reduce(const DevLayout& D, uint32_t rows, const uint32_t* partial, uint32_t* sums, uint32_t* enc, const uint64_t* out_off, uint8_t* out, hipStream_t st) {
    ReduceView R; R.rows = rows; R.ntargets = D.ntargets; R.partial = partial; R.target_chunk_begin = D.target_chunk_begin;
    R.enc = enc; R.out_off = out_off; R.out = out; R.corr = nullptr;
    ZKP_TRACED("k_sum_t<EdMsm>", st, launch_sum_ed(R, sums, st));
    ZKP_TRACED("k_encode", st, k_encode<<<dim3((rows + TW - 1) / TW, D.ntargets), TW, 0, st>>>(R, sums));
    return 0;
}
int msm_and_encode(const LayoutSet& S, uint32_t rows, const uint32_t* digits, uint32_t* partial, uint32_t* sums, uint32_t* enc, const uint64_t* out_off, uint8_t* out, hipStream_t st) {
    const DevLayout& D = pick_layout(S, rows);
    int rc = launch_msm(D, rows, digits, partial, st);
    if (rc) return rc;
    return launch_reduce(D, rows, partial, sums, enc, out_off, out, st);
}

// the whole prover for M jobs + C commitment tasks already described in the workspace
int run_pipeline(const Ws& w, uint32_t M, uint32_t C, hipStream_t st, SubBatch& lane) {
    int rc;
    const dim3 gj((M + TB - 1) / TB), gw((M + TW - 1) / TW);
    const uint32_t n = w.V.n, lg = w.V.lg;
    const Device::Family& F = dev().fam[lg - 3];
    // The commitment tasks (one small MSM, one inverse-square-root chain on C/64 waves) depend on nothing the proofs compute:
    // they go to a side stream and overlap the tape / first MSM instead of standing in front of them.
    const bool forked = C != 0 && M != 0;
    if (C) {
        hipStream_t cs = forked ? lane.side : st;
        if (forked) { HIP_TRY(hipEventRecord(lane.side_go, st)); HIP_TRY(hipStreamWaitEvent(cs, lane.side_go, 0)); }
        ZKP_TRACED("k_ctask", cs, k_ctask<<<(C + TB - 1) / TB, TB, 0, cs>>>(w.T));
        if ((rc = msm_and_encode(dev().ct, C, w.T.digits, w.ct_partial, w.ct_sums, w.ct_enc, w.ct_off, w.V.out, cs))) return rc;
        if (forked) HIP_TRY(hipEventRecord(lane.side_done, cs));
    }
    if (M == 0) { HIP_TRY(hipGetLastError()); return 0; }
    ZKP_TRACED("k_tape", st, k_tape<<<dim3(gj.x, tape_slots(n) + 1), TB, 0, st>>>(w.V));
    if ((rc = msm_and_encode(F.p1, M, w.V.d1, w.partial, w.sums, w.V.enc, nullptr, nullptr, st))) return rc;
    ZKP_TRACED("k_transcript1", st, k_transcript1<<<gw, TW, 0, st>>>(w.V));
    ZKP_TRACED("k_poly", st, k_poly<<<dim3(gj.x, n), TB, 0, st>>>(w.V));
    ZKP_TRACED("k_poly_sum", st, k_poly_sum<<<(M + 7) / 8, TW, 0, st>>>(w.V));
    if ((rc = msm_and_encode(dev().p2, M, w.V.d2, w.partial, w.sums, w.V.enc, nullptr, nullptr, st))) return rc;
    ZKP_TRACED("k_transcript2", st, k_transcript2<<<gw, TW, 0, st>>>(w.V));
    ZKP_TRACED("k_lr_init", st, k_lr_init<<<dim3(gj.x, n), TB, 0, st>>>(w.V));
    for (uint32_t r = 0; r < lg; r++) {
        ZKP_TRACED("k_round_prep", st, k_round_prep<<<dim3(gj.x, n), TB, 0, st>>>(w.V, r));
        ZKP_TRACED("k_round_sum", st, k_round_sum<<<gw, TW, 0, st>>>(w.V, r));
        if ((rc = msm_and_encode(F.rd[r], M, w.V.dr, w.partial, w.sums, w.V.enc, nullptr, nullptr, st))) return rc;
        ZKP_TRACED("k_transcript_round", st, k_transcript_round<<<gw, TW, 0, st>>>(w.V, r));
    }
    if (forked) HIP_TRY(hipStreamWaitEvent(st, lane.side_done, 0));
    HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void k_any_failed(const int32_t* status, uint32_t n, int* flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && status[i] != 0) atomicOr(flag, 1);
}

// Ops are independent, so the batch is cut into contiguous sub-batches that run the whole kernel sequence on their
// own streams: while one sub-batch is in a latency-bound per-proof step (transcript, inversion, encoding) the other's
// MSM keeps the CUs busy.  `st` (the caller's stream) is forked into the sub-streams and joined again.
int prove_range_device_locked(uint64_t n, const uint64_t* d_value, const uint64_t* d_min, const uint64_t* d_max, uint32_t lg,
                              const uint8_t* d_seeds, uint8_t* d_out, uint64_t stride, uint32_t* d_out_len, int32_t* d_status,
                              hipStream_t st, int* any_failed, int slot_hint = -1, bool masked = false) {
    if (n == 0) { if (any_failed) *any_failed = 0; return 0; }
    { int rcn = check_batch_size(n); if (rcn) return rcn; }
    if (stride < range_envelope_bytes(lg)) return fail(ZKP_HIP_E_ARGUMENT, "stride is smaller than the proof (1478 bytes for n_bits = 64)");
    int rc;
    if ((rc = ensure_family(lg))) return rc;
    uint32_t nsub = dev().nsub;
    if (n < 512) nsub = 1;                       // small batches: one stream
    const uint64_t per = (n + nsub - 1) / nsub;
    // slot (streams + workspace): the scheduler's lane when it calls; otherwise slot 0 unless a batch is still running there (a
    // caller that keeps two batches in flight from two of its own streams then gets both slots)
    uint32_t slot = 0;
    if (slot_hint >= 0) slot = (uint32_t)slot_hint % NSLOTS;
    else {
        SubBatch& s0 = dev().sub[0];
        if (s0.used && hipEventQuery(s0.done) == hipErrorNotReady) { slot = 1 + dev().next_slot % (NSLOTS - 1); dev().next_slot++; }
        (void)hipGetLastError();
    }
    if (masked) {                                  // the scheduler's CU partition: one slice on the lane's masked streams
        nsub = 1;
        if (dev().subm.size() < NSLOTS) dev().subm.resize(NSLOTS);
        if ((rc = ensure_sub(dev().subm[slot], true, &dev().subm[0]))) return rc;
    } else for (uint32_t h = 0; h < dev().nsub; h++) if ((rc = ensure_sub(dev().sub[(size_t)slot * dev().nsub + h], false, &dev().sub[h]))) return rc;
    SubBatch& first = masked ? dev().subm[slot] : dev().sub[(size_t)slot * dev().nsub];
    HIP_TRY(hipEventRecord(first.start, st));
    for (uint32_t h = 0; h < nsub; h++) {
        const uint64_t lo = h * per, hi = (lo + per < n) ? lo + per : n;
        if (lo >= hi) continue;
        SubBatch& sb = masked ? dev().subm[slot] : dev().sub[(size_t)slot * dev().nsub + h];
        const uint32_t C = (uint32_t)(hi - lo), M = 2 * C;
        if ((rc = ensure_workspace(sb, M, C))) return rc;
        Ws w; carve((uint8_t*)sb.ws, M, C, dev().max_chunks, &w);
        w.V.n = 1u << lg; w.V.lg = lg;
        w.V.seeds = reinterpret_cast<const uint32_t*>(d_seeds + 32 * lo); w.T.seeds = w.V.seeds;
        w.V.out = d_out + lo * stride;
        HIP_TRY(hipStreamWaitEvent(sb.stream, first.start, 0));        // (the slot's previous batch is ahead of this one on sb.stream)
        if (sb.used) HIP_TRY(hipStreamWaitEvent(sb.stream, sb.done, 0));   // ... unless it was a host-described job list on another stream (run_jobs_on)
        ZKP_TRACED("k_build_range", sb.stream, k_build_range<<<(C + TB - 1) / TB, TB, 0, sb.stream>>>(w.J, C, d_value + lo, d_min + lo, d_max + lo, lg, w.V.out, stride, d_out_len + lo, d_status + lo));
        if ((rc = run_pipeline(w, M, C, sb.stream, sb))) return rc;
        HIP_TRY(hipEventRecord(sb.done, sb.stream)); sb.used = true;
        HIP_TRY(hipStreamWaitEvent(st, sb.done, 0));
    }
    if (any_failed) {
        Ws w; carve((uint8_t*)first.ws, first.capM, first.capC, dev().max_chunks, &w);
        HIP_TRY(hipMemsetAsync(w.flag, 0, sizeof(int), st));
        k_any_failed<<<(uint32_t)((n + TB - 1) / TB), TB, 0, st>>>(d_status, (uint32_t)n, w.flag);
        HIP_TRY(hipMemcpyAsync(any_failed, w.flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Host-described jobs (threshold / consistency framings have variable-length inputs, so their job lists are built on
// the host: pure bookkeeping, bulletproofs.rs:309-437).  The proving itself is the same device pipeline.
struct HostJobs {
    std::vector<uint64_t> v, proof_off, commit_off, ct_v, ct_off;
    std::vector<uint32_t> seed_ix, proof_ix, ct_seed_ix, ct_bl_ix;
    std::vector<int32_t> bl_plus, bl_minus;
    std::vector<uint8_t> kind;
    void add_job(uint64_t val, uint32_t seed, uint32_t pidx, int32_t plus, int32_t minus, uint8_t k, uint64_t poff, uint64_t coff) {
        v.push_back(val); seed_ix.push_back(seed); proof_ix.push_back(pidx); bl_plus.push_back(plus); bl_minus.push_back(minus);
        kind.push_back(k); proof_off.push_back(poff); commit_off.push_back(coff);
    }
    void add_commit(uint64_t val, uint32_t seed, uint32_t bl, uint64_t off) { ct_v.push_back(val); ct_seed_ix.push_back(seed); ct_bl_ix.push_back(bl); ct_off.push_back(off); }
};

// Uploads a host-built job list into the workspace of `sb` and runs the prover on `st`.  d_img: device image that already
// holds every framing byte (proof_off / commit_off / ct_off are offsets into it); d_seeds: 32 bytes per seed index.
// The job arrays of H must stay alive until `st` has passed the copies (callers keep H until they synchronise).
int run_jobs_on(SubBatch& sb, const HostJobs& H, const uint8_t* d_seeds, uint8_t* d_img, uint32_t lg, hipStream_t st) {
    const uint32_t M = (uint32_t)H.v.size(), C = (uint32_t)H.ct_v.size();
    if (M == 0 && C == 0) return 0;
    int rc;
    if ((rc = ensure_family(lg)) || (rc = ensure_sub(sb))) return rc;
    if (sb.used) HIP_TRY(hipStreamWaitEvent(st, sb.done, 0));          // an asynchronous device-pointer call may still own this workspace
    if ((rc = ensure_workspace(sb, M ? M : 1, C ? C : 1))) return rc;
    Ws w; carve((uint8_t*)sb.ws, M ? M : 1, C ? C : 1, dev().max_chunks, &w);
    w.V.M = M; w.T.C = C; w.V.n = 1u << lg; w.V.lg = lg;
#define UP(dst, vec) do { if (!(vec).empty()) HIP_TRY(hipMemcpyAsync((void*)(dst), (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, st)); } while (0)
    UP(w.J.v, H.v); UP(w.J.seed_ix, H.seed_ix); UP(w.J.proof_ix, H.proof_ix); UP(w.J.bl_plus, H.bl_plus); UP(w.J.bl_minus, H.bl_minus);
    UP(w.J.kind, H.kind); UP(w.J.proof_off, H.proof_off); UP(w.J.commit_off, H.commit_off);
    UP(w.J.ct_v, H.ct_v); UP(w.J.ct_seed_ix, H.ct_seed_ix); UP(w.J.ct_bl_ix, H.ct_bl_ix); UP(w.J.ct_off, H.ct_off);
#undef UP
    w.V.seeds = reinterpret_cast<const uint32_t*>(d_seeds); w.T.seeds = w.V.seeds; w.V.out = d_img;
    if ((rc = run_pipeline(w, M, C, st, sb))) return rc;
    HIP_TRY(hipEventRecord(sb.done, st)); sb.used = true;
    return 0;
}

// out: host buffer already holding every framing byte; proofs and commitments are filled in by the device
int run_host_jobs(const HostJobs& H, const uint8_t* seeds, size_t nseeds, uint8_t* out, size_t out_bytes, uint32_t lg = 6) {
    if (H.v.empty() && H.ct_v.empty()) return 0;
    hipStream_t st = dev().stream;
    DevScope mem;
    uint8_t *d_seeds = nullptr, *d_out = nullptr;
    HIP_TRY(mem.alloc(&d_seeds, 32 * nseeds)); HIP_TRY(mem.alloc(&d_out, out_bytes));
    HIP_TRY(hipMemcpyAsync(d_seeds, seeds, 32 * nseeds, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_out, out, out_bytes, hipMemcpyHostToDevice, st));
    int rc = run_jobs_on(dev().sub[0], H, d_seeds, d_out, lg, st);
    if (rc == 0) HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));          // nothing may still be using the per-call buffers when `mem` goes
    return rc;
}

void put_le_host(uint8_t* p, uint64_t x, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)(x >> (8 * i)); }

int fresh_seeds(std::vector<uint8_t>& buf, size_t n) {
    buf.resize(32 * n);
    size_t got = 0;
    while (got < buf.size()) {
        ssize_t r = getrandom(buf.data() + got, buf.size() - got, 0);
        if (r <= 0) return fail(ZKP_HIP_E_RUNTIME, "getrandom failed");
        got += (size_t)r;
    }
    return 0;
}

uint64_t threshold_envelope_bytes(uint32_t lg) { return 10 + 8 + 4 + 4 + rp_bytes(lg) + 32 + 32; }      // 762 for n_bits = 64
uint64_t consistency_envelope_bytes(uint32_t count) {
    if (count == 0) return 0;
    return 10 + 4 + 32ull * count + (uint64_t)(4 + RP_BYTES) * (count - 1) + 32ull * (count - 1) + 32;
}
// Framing of threshold proofs (threshold_proof.rs:12-32 -> bulletproofs.rs:309-366): validates every op, writes the envelope
// and body framing of the valid ones into `img` (n records of `stride` bytes, zeroed by the caller) and appends one proof
// job + one commitment task per valid op; seed index = op index.  Returns 1 if any op is invalid.
int frame_threshold(uint64_t n, const uint64_t* values, const uint32_t* counts, const uint64_t* thresholds, uint32_t lg,
                    uint8_t* img, uint64_t stride, uint32_t* out_len, int32_t* status, HostJobs& H) {
    const uint32_t RP = rp_bytes(lg), n_bits = 1u << lg;
    const uint64_t PB = threshold_envelope_bytes(lg);
    const uint64_t max_diff = lg >= 6 ? ~0ull : (1ull << (1u << lg)) - 1;   // bulletproofs.rs:330-336
    int any = 0; size_t pos = 0;
    for (uint64_t i = 0; i < n; i++) {
        // validation.rs:30-47 / bulletproofs.rs:314-336
        bool ok = counts[i] > 0; uint64_t sum = 0;
        for (uint32_t k = 0; k < counts[i] && ok; k++) { const uint64_t x = values[pos + k]; if (sum + x < sum) ok = false; sum += x; }
        pos += counts[i];
        if (ok && sum < thresholds[i]) ok = false;
        if (ok && sum - thresholds[i] > max_diff) ok = false;
        status[i] = ok ? ZKP_HIP_OK : ZKP_HIP_INVALID_INPUT; out_len[i] = ok ? (uint32_t)PB : 0; any |= !ok;
        if (!ok) continue;
        uint8_t* o = img + i * stride; const uint64_t base = i * stride;
        o[0] = 2; o[1] = 3; put_le_host(o + 2, 8 + 4 + 4 + RP + 32, 4); put_le_host(o + 6, 32, 4);
        put_le_host(o + 10, thresholds[i], 8); put_le_host(o + 18, n_bits, 4); put_le_host(o + 22, RP, 4);
        H.add_job(sum - thresholds[i], (uint32_t)i, 0, 0, -1, KIND_THRESHOLD, base + 26, base + 26 + RP);
        H.add_commit(sum, (uint32_t)i, 0, base + 26 + RP + 32);
    }
    return any;
}
// Framing of consistency proofs (consistency_proof.rs:12-22 -> bulletproofs.rs:368-437): k commitment tasks and k - 1 proof
// jobs per valid op.  The envelope's commitment field (SHA-256 of the commitment list) is filled in after the device run.
int frame_consistency(uint64_t n, const uint64_t* data, const uint32_t* counts, uint8_t* img, uint64_t stride, uint32_t* out_len, int32_t* status, HostJobs& H) {
    int any = 0; size_t pos = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t k = counts[i]; const uint64_t* d = data + pos; pos += k;
        bool ok = k > 0;                                            // validation.rs:75-88
        for (uint32_t j = 1; j < k && ok; j++) if (d[j - 1] > d[j]) ok = false;
        const uint64_t need = consistency_envelope_bytes(k);
        if (ok && need > stride) ok = false;                        // (callers size the stride from the counts; never reached through the ABI)
        status[i] = ok ? ZKP_HIP_OK : ZKP_HIP_INVALID_INPUT; out_len[i] = ok ? (uint32_t)need : 0; any |= !ok;
        if (!ok) continue;
        uint8_t* o = img + i * stride; const uint64_t base = i * stride;
        const uint64_t body = need - 10 - 32;
        o[0] = 2; o[1] = 6; put_le_host(o + 2, body, 4); put_le_host(o + 6, 32, 4);
        put_le_host(o + 10, k, 4);
        const uint64_t commits = base + 14, proofs = commits + 32ull * k, dcs = proofs + (uint64_t)(4 + RP_BYTES) * (k - 1);
        for (uint32_t j = 0; j < k; j++) H.add_commit(d[j], (uint32_t)i, j, commits + 32ull * j);
        for (uint32_t j = 1; j < k; j++) {
            put_le_host(img + proofs + (uint64_t)(4 + RP_BYTES) * (j - 1), RP_BYTES, 4);
            H.add_job(d[j] - d[j - 1], (uint32_t)i, j - 1, (int32_t)j, (int32_t)(j - 1), KIND_CONSISTENCY,
                      proofs + (uint64_t)(4 + RP_BYTES) * (j - 1) + 4, dcs + 32ull * (j - 1));
        }
    }
    return any;
}

}  // namespace

#include "g16_impl.inc"
#include "stark_impl.inc"
#include "bpv_impl.inc"

// ================================================================================================ C ABI
extern "C" {

const char* zkp_hip_last_error(void) { return t_err.c_str(); }
void zkp_hip_set_window_budget(uint32_t budget) try { g_budget_request = budget; } ZKP_API_CATCH_VOID
void zkp_hip_set_subbatches(uint32_t n) try { g_subbatches = n; } ZKP_API_CATCH_VOID
void zkp_hip_set_msm_variant(uint32_t v) try { if (v >= 100) g_fill = v / 100.0; } ZKP_API_CATCH_VOID   // benchmarking knob: grid fill target x100 (single kernel variant remains)

int zkp_hip_init(int device) try {
    Device* d = nullptr;
    {
        Registry& R = registry();
        std::lock_guard<std::mutex> lk(R.mu);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(ZKP_HIP_E_RUNTIME, "zkp_hip_init: no HIP device available (this library has no CPU fallback)");
        if (device < 0 || device >= ndev) return fail(ZKP_HIP_E_ARGUMENT, "zkp_hip_init: bad device index");
        for (Device* s : R.shards) if (s->hip_dev == device) { d = s; break; }
        if (!d) { d = new Device(); d->index = (int)R.shards.size(); d->hip_dev = device; R.shards.push_back(d); }
        if (!R.hooked) { R.hooked = true; (void)atexit(zkp_hip_shutdown); }
    }
    Bind bind; int rc = bind.open(d);
    return rc;
} ZKP_API_CATCH_INT

int zkp_hip_init_devices(uint32_t count, const int* devices) try {
    if (count == 0 || count > 64 || !devices) return fail(ZKP_HIP_E_ARGUMENT, "zkp_hip_init_devices: 1..64 shards");
    std::vector<Device*> mine;
    {
        Registry& R = registry();
        std::lock_guard<std::mutex> lk(R.mu);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(ZKP_HIP_E_RUNTIME, "zkp_hip_init: no HIP device available (this library has no CPU fallback)");
        for (uint32_t k = 0; k < count; k++) if (devices[k] < 0 || devices[k] >= ndev) return fail(ZKP_HIP_E_ARGUMENT, "zkp_hip_init_devices: bad device index");
        if (!R.shards.empty()) {                       // idempotent for the same assignment; anything else needs a shutdown first
            bool same = R.shards.size() == count;
            for (uint32_t k = 0; same && k < count; k++) same = R.shards[k]->hip_dev == devices[k];
            if (!same) return fail(ZKP_HIP_E_ARGUMENT, "zkp_hip_init_devices: shards are already registered differently (zkp_hip_shutdown first)");
        } else {
            for (uint32_t k = 0; k < count; k++) { Device* d = new Device(); d->index = (int)k; d->hip_dev = devices[k]; R.shards.push_back(d); }
        }
        mine = R.shards;
        if (!R.hooked) { R.hooked = true; (void)atexit(zkp_hip_shutdown); }
    }
    std::vector<int> rcs(count, 0); std::vector<std::string> errs(count);
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < count; k++) th.emplace_back([&, k]() { rcs[k] = guarded([&]() { Bind bind; return bind.open(mine[k]); }); if (rcs[k]) { try { errs[k] = t_err; } catch (...) {} } });
    for (auto& t : th) t.join();
    for (uint32_t k = 0; k < count; k++) if (rcs[k]) return fail(rcs[k], errs[k]);
    return 0;
} ZKP_API_CATCH_INT

int zkp_hip_device_count(void) try { Registry& R = registry(); std::lock_guard<std::mutex> lk(R.mu); return (int)R.shards.size(); } ZKP_API_CATCH_INT

int zkp_hip_use_device(int shard) try {
    Registry& R = registry(); std::lock_guard<std::mutex> lk(R.mu);
    if (shard < 0 || (shard > 0 && shard >= (int)R.shards.size())) return fail(ZKP_HIP_E_ARGUMENT, "zkp_hip_use_device: no such shard");
    t_sel = shard;
    return 0;
} ZKP_API_CATCH_INT

// Releases every device resource of every shard and forgets the shard registration.  Also the library's atexit hook: it
// runs while the HIP runtime is still alive, so nothing of ours (streams, events, allocations, code objects in use) is
// left for the runtime's own exit-time teardown to trip over.  The Device objects themselves are never freed.
void zkp_hip_shutdown(void) try {
    std::vector<Device*> shards;
    { Registry& R = registry(); std::lock_guard<std::mutex> lk(R.mu); shards.swap(R.shards); }
    t_sel = 0;
    for (Device* d : shards) {
        std::lock_guard<std::mutex> lk(d->mu);
        if (!d->ready) continue;
        Device* prev = t_dev; t_dev = d;
        (void)hipSetDevice(d->hip_dev);
        (void)hipDeviceSynchronize();
        batch_release_all();
        g16_release_all();
        bpv_release_all();
        stark_release_all();
        d->pool.release_all();
        for (auto* vec : {&d->sub, &d->subm}) {
            for (auto& sb : *vec) {
                if (sb.ws) (void)hipFree(sb.ws);
                if (!sb.stream) continue;
                if (!sb.borrowed) { (void)hipStreamDestroy(sb.stream); (void)hipStreamDestroy(sb.side); }
                (void)hipEventDestroy(sb.start); (void)hipEventDestroy(sb.done); (void)hipEventDestroy(sb.side_go); (void)hipEventDestroy(sb.side_done);
            }
            vec->clear();
        }
        d->cus_now = 0;
        d->msm_prio_now = 0;
        release_edg_table();
        free_set(d->p2); free_set(d->ct);
        for (auto& F : d->fam) { free_set(F.p1); for (auto& s : F.rd) free_set(s); F.ready = false; }
        trace_release();
        for (auto& K : d->prof) { for (auto& e : K.ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } K = Device::KProf(); }
        (void)hipStreamDestroy(d->stream); d->stream = nullptr;
        d->ready = false; d->profiling = false; d->max_chunks = 0; d->next_slot = 0; d->generation++;
        t_dev = prev;
    }
} ZKP_API_CATCH_VOID

void zkp_hip_profile_enable(int on) try {
    std::vector<Device*> shards;      // (the registry lock is not held while a shard's is taken: Bind::open takes them in the other order)
    { Registry& R = registry(); std::lock_guard<std::mutex> lk(R.mu); shards = R.shards; }
    for (Device* d : shards) { std::lock_guard<std::mutex> dl(d->mu); d->profiling = on != 0; }
} ZKP_API_CATCH_VOID

// accumulated over all shards
int zkp_hip_profile_read_kernel(int which, double* ms, uint64_t* launches, uint64_t* point_adds, int reset) try {
    if (which < 0 || which > 2) return fail(ZKP_HIP_E_ARGUMENT, "unknown kernel id");
    std::vector<Device*> shards;
    { Registry& R = registry(); std::lock_guard<std::mutex> lk(R.mu); shards = R.shards; }
    double tms = 0; uint64_t tl = 0, ta = 0;
    for (Device* d : shards) {
        Bind bind; int rc = bind.open(d);
        if (rc) return rc;
        HIP_TRY(hipDeviceSynchronize());
        Device::KProf& K = d->prof[which];
        for (size_t i = 0; i < K.ev_used; i++) { float t = 0; HIP_TRY(hipEventElapsedTime(&t, K.ev_pool[i].first, K.ev_pool[i].second)); K.ms += t; }
        K.ev_used = 0;
        tms += K.ms; tl += K.launches; ta += K.adds;
        if (reset) { K.ms = 0; K.launches = 0; K.adds = 0; }
    }
    if (ms) *ms = tms;
    if (launches) *launches = tl;
    if (point_adds) *point_adds = ta;
    return 0;
} ZKP_API_CATCH_INT
int zkp_hip_profile_read(double* msm_ms, uint64_t* msm_launches, uint64_t* msm_point_adds, int reset) try {
    return zkp_hip_profile_read_kernel(0, msm_ms, msm_launches, msm_point_adds, reset);
} ZKP_API_CATCH_INT

int zkp_hip_prove_range_batch_device(uint64_t n, const uint64_t* d_value, const uint64_t* d_min, const uint64_t* d_max, uint32_t n_bits,
                                     const uint8_t* d_seeds, uint8_t* d_out, uint64_t stride, uint32_t* d_out_len, int32_t* d_status,
                                     void* stream, int* any_failed) try {
    uint32_t lg;
    if (!bits_to_lg(n_bits, &lg)) return fail(ZKP_HIP_E_UNSUPPORTED, "n_bits must be 8, 16, 32 or 64");
    if (!d_seeds) return fail(ZKP_HIP_E_ARGUMENT, "device entry point needs seeds");
    Bind bind; int rc = bind.open();
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : dev().stream;
    rc = prove_range_device_locked(n, d_value, d_min, d_max, lg, d_seeds, d_out, stride, d_out_len, d_status, st, any_failed);
    if (rc) return rc;
    return (any_failed && *any_failed) ? 1 : 0;
} ZKP_API_CATCH_INT

int zkp_hip_prove_range_batch(uint64_t n, const uint64_t* value, const uint64_t* min, const uint64_t* max, uint32_t n_bits,
                              const uint8_t* seeds, uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status) try {
    uint32_t lg;
    if (!bits_to_lg(n_bits, &lg)) return fail(ZKP_HIP_E_UNSUPPORTED, "n_bits must be 8, 16, 32 or 64");
    if (n == 0) return 0;
    if (!value || !min || !max || !out || !out_len || !status) return fail(ZKP_HIP_E_ARGUMENT, "null pointer argument");
    { int rcn = check_batch_size(n); if (rcn) return rcn; }
    if (stride < range_envelope_bytes(lg)) return fail(ZKP_HIP_E_ARGUMENT, "stride is smaller than the proof (1478 bytes for n_bits = 64)");
    std::vector<uint8_t> fresh;
    if (!seeds) { int rc0 = fresh_seeds(fresh, n); if (rc0) return rc0; seeds = fresh.data(); }   // bulletproofs.rs:82-87
    Bind bind; int rc = bind.open();
    if (rc) return rc;
    hipStream_t st = dev().stream;
    DevScope mem;
    uint64_t *d_in = nullptr; uint8_t *d_seeds = nullptr, *d_out = nullptr; uint32_t* d_len = nullptr; int32_t* d_status = nullptr;
    HIP_TRY(mem.alloc(&d_in, 24 * n)); HIP_TRY(mem.alloc(&d_seeds, 32 * n)); HIP_TRY(mem.alloc(&d_out, stride * n));
    HIP_TRY(mem.alloc(&d_len, 4 * n)); HIP_TRY(mem.alloc(&d_status, 4 * n));
    trace_origin(st);
    HIP_TRY(hipMemcpyAsync(d_in, value, 8 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_in + n, min, 8 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_in + 2 * n, max, 8 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_seeds, seeds, 32 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_out, 0, stride * n, st));
    int any = 0;
    rc = prove_range_device_locked(n, d_in, d_in + n, d_in + 2 * n, lg, d_seeds, d_out, stride, d_len, d_status, st, &any);
    if (rc == 0) {
        HIP_TRY(hipMemcpyAsync(out, d_out, stride * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_len, d_len, 4 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(status, d_status, 4 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        trace_dump();
        // a failed item leaves no proof bytes behind (reference returns Err, never a partial Vec)
        for (uint64_t i = 0; i < n; i++) if (status[i] != 0) memset(out + i * stride, 0, stride);
    }
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    return any ? 1 : 0;
} ZKP_API_CATCH_INT

uint64_t zkp_hip_range_proof_bytes(uint32_t n_bits) try { uint32_t lg; return bits_to_lg(n_bits, &lg) ? range_envelope_bytes(lg) : 0; } ZKP_API_CATCH_ZERO
uint64_t zkp_hip_threshold_proof_bytes(uint32_t n_bits) try { uint32_t lg; return bits_to_lg(n_bits, &lg) ? threshold_envelope_bytes(lg) : 0; } ZKP_API_CATCH_ZERO
uint64_t zkp_hip_consistency_proof_bytes(uint32_t count) try { return consistency_envelope_bytes(count); } ZKP_API_CATCH_ZERO

int zkp_hip_prove_threshold_batch(uint64_t n, const uint64_t* values, const uint32_t* counts, const uint64_t* thresholds, uint32_t n_bits,
                                  const uint8_t* seeds, uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status) try {
    uint32_t lg;
    if (!bits_to_lg(n_bits, &lg)) return fail(ZKP_HIP_E_UNSUPPORTED, "n_bits must be 8, 16, 32 or 64");
    if (n == 0) return 0;
    if (!values || !counts || !thresholds || !out || !out_len || !status) return fail(ZKP_HIP_E_ARGUMENT, "null pointer argument");
    { int rcn = check_batch_size(n); if (rcn || (rcn = check_list_total(n, counts))) return rcn; }
    if (stride < threshold_envelope_bytes(lg)) return fail(ZKP_HIP_E_ARGUMENT, "stride is smaller than the proof (762 bytes for n_bits = 64)");
    std::vector<uint8_t> fresh;
    if (!seeds) { int rc = fresh_seeds(fresh, n); if (rc) return rc; seeds = fresh.data(); }
    HostJobs H;
    memset(out, 0, stride * n);
    const int any = frame_threshold(n, values, counts, thresholds, lg, out, stride, out_len, status, H);
    Bind bind; int rc = bind.open();
    if (rc) return rc;
    if ((rc = run_host_jobs(H, seeds, n, out, stride * n, lg))) return rc;
    return any;
} ZKP_API_CATCH_INT

int zkp_hip_prove_consistency_batch(uint64_t n, const uint64_t* data, const uint32_t* counts, const uint8_t* seeds,
                                    uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status) try {
    if (n == 0) return 0;
    if (!data || !counts || !out || !out_len || !status) return fail(ZKP_HIP_E_ARGUMENT, "null pointer argument");
    { int rcn = check_batch_size(n); if (rcn || (rcn = check_list_total(n, counts))) return rcn; }
    std::vector<uint8_t> fresh;
    if (!seeds) { int rc = fresh_seeds(fresh, n); if (rc) return rc; seeds = fresh.data(); }
    size_t pos = 0;
    for (uint64_t i = 0; i < n; i++) {
        bool ok = counts[i] > 0; for (uint32_t j = 1; j < counts[i] && ok; j++) if (data[pos + j - 1] > data[pos + j]) ok = false;
        if (ok && zkp_hip_consistency_proof_bytes(counts[i]) > stride) return fail(ZKP_HIP_E_ARGUMENT, "stride too small for a consistency proof (see zkp_hip_consistency_proof_bytes)");
        pos += counts[i];
    }
    HostJobs H;
    memset(out, 0, stride * n);
    const int any = frame_consistency(n, data, counts, out, stride, out_len, status, H);
    {
        Bind bind; int rc = bind.open();
        if (rc) return rc;
        if ((rc = run_host_jobs(H, seeds, n, out, stride * n))) return rc;
    }
    for (uint64_t i = 0; i < n; i++) {          // commitment field = SHA-256 of the commitment list (bulletproofs.rs:430-436)
        if (status[i] != 0) continue;
        uint8_t* o = out + i * stride;
        sha256_host(o + out_len[i] - 32, o + 14, 32ull * counts[i]);
    }
    return any;
} ZKP_API_CATCH_INT

}  // extern "C"

#include "batch_impl.inc"
