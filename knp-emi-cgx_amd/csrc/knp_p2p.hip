// Native peer-to-peer exchange for the multi-GPU path (halo of the Krylov / level vectors before each
// SpMV, SUM of the inner-product partials).  Replaces the per-exchange Python hook + RCCL call by ONE kernel launch:
//
//   phase 1 (send)     gathers x[send_idx] and stores it straight into the receiver's mailbox (uncached device memory of
//                      the peer, mapped with hipIpcOpenMemHandle -> xGMI stores between GPUs); the last block to finish
//                      publishes the message's sequence number to the receiver's flag
//   phase 2 (receive)  thread 0 of each block waits for the peers' flags (acquire, system scope, bounded by a wall-clock
//                      timeout), then the block copies / adds its own mailbox into the vector
//
// Mailboxes are double buffered by the parity of the sequence number.  Because every neighbour relation is symmetric
// (both directions are always signalled, also with zero entries) message q+2 cannot overtake the consumption of
// message q: the sender's own receive phase of q+1 needed the receiver's send phase of q+1, which the receiver's
// stream orders after its receive phase of q.  The same argument covers the all-reduce (everybody posts to everybody).
// The all-reduce sums in rank order, so all ranks get bit-identical results (they take the same convergence decisions).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "knp_internal.hpp"

#define P2P_NT 256
#define P2P_FLAG_STRIDE 64   // bytes between flags

#define PCHK(call)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            ctx->err = std::string("HIP error (p2p): ") + hipGetErrorString(e_) + " at " #call;     \
            return KNP_E_HIP;                                                                        \
        }                                                                                            \
    } while (0)

struct P2PSend {
    int n;
    int64_t ptr[KNP_P2P_MAXPEERS + 1];
    double* dst[KNP_P2P_MAXPEERS];
    int64_t* flag[KNP_P2P_MAXPEERS];
};
struct P2PWait {
    int n;
    const int64_t* flag[KNP_P2P_MAXPEERS];
};

// err lives in pinned host memory (the host polls it), d_err is its twin in device memory: once a
// wait has timed out every later wait gives up at once, so a dead peer costs one timeout, not one per exchange
__device__ __forceinline__ void p2p_wait(const int64_t* flag, int64_t seq, int64_t timeout_ticks, int* err, const int* d_err) {
    const int64_t t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(8);
        const bool gave_up = __hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (gave_up || (int64_t)wall_clock64() - t0 > timeout_ticks) {   // the peer never arrived: report and go on (never hang the GPU)
            __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_or(const_cast<int*>(d_err), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// after this block's stores: the last block to arrive publishes seq to every destination flag
__device__ __forceinline__ void p2p_publish(const P2PSend& a, unsigned int* counter, int64_t seq) {
    __shared__ int last;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(counter, 1u) == gridDim.x - 1);
    __syncthreads();
    if (last) {
        __threadfence_system();
        if ((int)threadIdx.x < a.n) __hip_atomic_store(a.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) *counter = 0u;
    }
}

// ---- one kernel per exchange -----------------------------------------------------------------------------------------
// Phase 1 writes this rank's data into the peers' mailboxes and (last block) publishes the sequence number; phase 2 waits
// for the peers' numbers and consumes the own mailbox.  A block may reach phase 2 before the own publish happened: it then
// only waits for PEERS, whose publish depends on their own phase 1 alone, so there is no circular wait as long as every
// block of the grid is resident.  The grid is sized from the message (one block per 1 K entries; a 3-4 MB halo of a
// 10^7-DoF subdomain gets the full 256 instead of the 64 blocks of the first version) and capped at
// one block per CU (256): with up to 6 ranks sharing one GPU (the test boxes) that is 6 blocks per CU, within the 8 that
// 256-thread blocks of this register footprint are admitted at, so every block of every rank is resident at the same time.
#define P2P_MAX_BLOCKS 256
#define P2P_ENTRIES_PER_BLOCK 1024

__device__ __forceinline__ void p2p_wait_all(const P2PWait& w, int64_t seq, int64_t timeout, int* err, const int* d_err) {
    if (threadIdx.x == 0)
        for (int j = 0; j < w.n; ++j) p2p_wait(w.flag[j], seq, timeout, err, d_err);
    __syncthreads();
}

// forward halo: ghost entries of x <- owners
__global__ void __launch_bounds__(P2P_NT) k_p2p_halo_fwd(P2PSend a, P2PWait w, const int32_t* __restrict__ send_idx, int64_t n_ghost,
                                                         const int32_t* __restrict__ recv_idx, const double* src, double* x,
                                                         unsigned int* counter, int64_t seq, int64_t timeout, int* err, const int* d_err) {
    const int64_t total = a.ptr[a.n];
    for (int64_t k = (int64_t)blockIdx.x * P2P_NT + threadIdx.x; k < total; k += (int64_t)gridDim.x * P2P_NT) {
        int j = 0;
        while (k >= a.ptr[j + 1]) ++j;
        a.dst[j][k - a.ptr[j]] = x[send_idx[k]];
    }
    p2p_publish(a, counter, seq);
    p2p_wait_all(w, seq, timeout, err, d_err);
    for (int64_t k = (int64_t)blockIdx.x * P2P_NT + threadIdx.x; k < n_ghost; k += (int64_t)gridDim.x * P2P_NT) x[recv_idx[k]] = src[k];
}

// reverse halo: owned entries of x += the ghost copies the peers hold.  Every owned entry sums its copies in peer order
// (dst / ptr / pos are built at connect time), so the result does not depend on the order of arrival.
__global__ void __launch_bounds__(P2P_NT) k_p2p_halo_rev(P2PSend a, P2PWait w, const int32_t* __restrict__ ghost_idx, int64_t n_dst,
                                                         const int32_t* __restrict__ dst, const int32_t* __restrict__ ptr,
                                                         const int32_t* __restrict__ pos, const double* src, double* x,
                                                         unsigned int* counter, int64_t seq, int64_t timeout, int* err, const int* d_err) {
    const int64_t total = a.ptr[a.n];
    for (int64_t k = (int64_t)blockIdx.x * P2P_NT + threadIdx.x; k < total; k += (int64_t)gridDim.x * P2P_NT) {
        int j = 0;
        while (k >= a.ptr[j + 1]) ++j;
        a.dst[j][k - a.ptr[j]] = x[ghost_idx[k]];
    }
    p2p_publish(a, counter, seq);
    p2p_wait_all(w, seq, timeout, err, d_err);
    for (int64_t i = (int64_t)blockIdx.x * P2P_NT + threadIdx.x; i < n_dst; i += (int64_t)gridDim.x * P2P_NT) {
        double s = 0.0;
        for (int q = ptr[i]; q < ptr[i + 1]; ++q) s += src[pos[q]];
        x[dst[i]] += s;
    }
}

// all-reduce: my vector -> slot [my rank] of everybody's mailbox (own included), then out = sum over ranks in rank order
// (bit-identical on all ranks), optionally mirrored to pinned host memory + sequence word (single-block launches).
// In place is fine: phase 2 starts after the own flag, i.e. after every block of this rank has read v.
__global__ void __launch_bounds__(P2P_NT) k_p2p_allreduce(P2PSend a, P2PWait w, int n, const double* v, int64_t stride, const double* src,
                                                          double* out, double* mirror, volatile int64_t* pub, int64_t pub_val,
                                                          unsigned int* counter, int64_t seq, int64_t timeout, int* err, const int* d_err) {
    const int64_t total = (int64_t)n * a.n;
    for (int64_t e = (int64_t)blockIdx.x * P2P_NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * P2P_NT) {
        const int r = (int)(e / n);
        const int k = (int)(e - (int64_t)r * n);
        a.dst[r][k] = v[k];
    }
    p2p_publish(a, counter, seq);
    p2p_wait_all(w, seq, timeout, err, d_err);
    for (int k = blockIdx.x * P2P_NT + threadIdx.x; k < n; k += gridDim.x * P2P_NT) {
        double s = 0.0;
        for (int r = 0; r < w.n; ++r) s += src[(int64_t)r * stride + k];
        out[k] = s;
        if (mirror) mirror[k] = s;
    }
    if (pub) {   // single-block launches only
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *pub = pub_val;
    }
}

static inline int p2p_blocks(int64_t n) {
    static const int cap = getenv("KNP_P2P_MAX_BLOCKS") ? std::max(1, atoi(getenv("KNP_P2P_MAX_BLOCKS"))) : P2P_MAX_BLOCKS;
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + P2P_ENTRIES_PER_BLOCK - 1) / P2P_ENTRIES_PER_BLOCK, cap));
}
static inline int64_t* flag_at(char* box, int which, int size, int r) {   // which: 0 forward / all-reduce, 1 reverse
    return reinterpret_cast<int64_t*>(box + ((size_t)which * size + r) * P2P_FLAG_STRIDE);
}

static KnpP2PPlan* get_plan(knp_ctx* ctx, int plan, int kind) {
    if (!ctx->p2p || plan < 0 || plan >= (int)ctx->p2p->plans.size()) { ctx->err = "bad p2p plan"; return nullptr; }
    KnpP2PPlan& P = ctx->p2p->plans[plan];
    if (!P.connected || P.kind != kind) { ctx->err = "p2p plan not connected / wrong kind"; return nullptr; }
    return &P;
}

// Mailbox layout of a halo plan (element offsets are in doubles after the header):
//   header   : forward flags [size] | reverse flags [size], one per source rank, 64 bytes apart
//   forward  : parity 0 [n_fwd] | parity 1 [n_fwd]     values of my ghost entries, in the order of my recv list
//   reverse  : parity 0 [n_rev] | parity 1 [n_rev]     ghost copies of my owned entries, in the order of my send list
// all-reduce plan: header flags [size] ; data parity 0 [size][n_max] | parity 1 [size][n_max]
static inline size_t hdr_bytes_of(int kind, int size) { return (size_t)(kind == KNP_P2P_HALO ? 2 : 1) * size * P2P_FLAG_STRIDE; }

int knp_p2p_halo_forward(knp_ctx* ctx, int plan, double* x) {
    KnpP2PPlan* Pp = get_plan(ctx, plan, KNP_P2P_HALO);
    if (!Pp) return KNP_E_STATE;
    KnpP2PPlan& P = *Pp;
    KnpP2P& C = *ctx->p2p;
    const int64_t seq = ++P.seq_fwd;
    const int par = (int)(seq & 1);
    P2PSend a;
    P2PWait w;
    a.n = w.n = P.n_peers;
    for (int j = 0; j <= P.n_peers; ++j) a.ptr[j] = P.send_ptr[j];
    for (int j = 0; j < P.n_peers; ++j) {
        char* pb = P.peer_box[P.peer_rank[j]];
        a.dst[j] = reinterpret_cast<double*>(pb + P.hdr_bytes) + P.remote_fwd_off[2 * j + par];
        a.flag[j] = flag_at(pb, 0, C.size, C.rank);
        w.flag[j] = flag_at(P.box, 0, C.size, P.peer_rank[j]);
    }
    const double* src = reinterpret_cast<const double*>(P.box + P.hdr_bytes) + (int64_t)par * P.n_fwd;
    hipLaunchKernelGGL(k_p2p_halo_fwd, dim3(p2p_blocks(std::max(P.send_ptr[P.n_peers], P.n_fwd))), dim3(P2P_NT), 0, ctx->stream, a, w,
                       P.d_send_idx, P.n_fwd, P.d_recv_idx, src, x, P.d_counter, seq, C.timeout_ticks, C.h_err_dev, C.d_err);
    PCHK(hipGetLastError());
    return KNP_OK;
}

int knp_p2p_halo_reverse(knp_ctx* ctx, int plan, double* x) {
    KnpP2PPlan* Pp = get_plan(ctx, plan, KNP_P2P_HALO);
    if (!Pp) return KNP_E_STATE;
    KnpP2PPlan& P = *Pp;
    KnpP2P& C = *ctx->p2p;
    const int64_t seq = ++P.seq_rev;
    const int par = (int)(seq & 1);
    P2PSend a;
    P2PWait w;
    a.n = w.n = P.n_peers;
    for (int j = 0; j <= P.n_peers; ++j) a.ptr[j] = P.recv_ptr[j];     // roles swapped: I send my ghost copies
    for (int j = 0; j < P.n_peers; ++j) {
        char* pb = P.peer_box[P.peer_rank[j]];
        a.dst[j] = reinterpret_cast<double*>(pb + P.hdr_bytes) + P.remote_rev_off[2 * j + par];
        a.flag[j] = flag_at(pb, 1, C.size, C.rank);
        w.flag[j] = flag_at(P.box, 1, C.size, P.peer_rank[j]);
    }
    const double* src = reinterpret_cast<const double*>(P.box + P.hdr_bytes) + 2 * P.n_fwd + (int64_t)par * P.n_rev;
    hipLaunchKernelGGL(k_p2p_halo_rev, dim3(p2p_blocks(std::max<int64_t>(P.recv_ptr[P.n_peers], P.n_rev_dst))), dim3(P2P_NT), 0, ctx->stream, a, w,
                       P.d_recv_idx, P.n_rev_dst, P.d_rev_dst, P.d_rev_ptr, P.d_rev_pos, src, x, P.d_counter + 1, seq, C.timeout_ticks,
                       C.h_err_dev, C.d_err);
    PCHK(hipGetLastError());
    return KNP_OK;
}

int knp_p2p_allreduce(knp_ctx* ctx, int plan, double* v, int n, double* mirror, int64_t* seq_dev, int64_t seq_val) {
    KnpP2PPlan* Pp = get_plan(ctx, plan, KNP_P2P_ALLREDUCE);
    if (!Pp) return KNP_E_STATE;
    KnpP2PPlan& P = *Pp;
    KnpP2P& C = *ctx->p2p;
    if (n < 0 || n > P.n_fwd) { ctx->err = "p2p all-reduce longer than the plan"; return KNP_E_ARG; }
    if (n == 0) return KNP_OK;
    const int64_t seq = ++P.seq_fwd;
    const int par = (int)(seq & 1);
    P2PSend a;
    P2PWait w;
    a.n = w.n = C.size;
    for (int r = 0; r < C.size; ++r) {
        char* pb = P.peer_box[r];
        a.dst[r] = reinterpret_cast<double*>(pb + P.hdr_bytes) + ((int64_t)par * C.size + C.rank) * P.n_fwd;
        a.flag[r] = flag_at(pb, 0, C.size, C.rank);
        w.flag[r] = flag_at(P.box, 0, C.size, r);
    }
    const double* src = reinterpret_cast<const double*>(P.box + P.hdr_bytes) + (int64_t)par * C.size * P.n_fwd;
    const bool pub = seq_dev != nullptr && n <= P2P_NT;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(pub ? 1 : p2p_blocks((int64_t)n * C.size)), dim3(P2P_NT), 0, ctx->stream, a, w, n, v, P.n_fwd, src,
                       v, mirror, pub ? seq_dev : nullptr, seq_val, P.d_counter, seq, C.timeout_ticks, C.h_err_dev, C.d_err);
    PCHK(hipGetLastError());
    return KNP_OK;
}

int knp_p2p_check(knp_ctx* ctx) {
    if (ctx->p2p && ctx->p2p->h_err && __atomic_load_n(ctx->p2p->h_err, __ATOMIC_ACQUIRE) != 0) {
        ctx->err = "p2p exchange timed out waiting for a peer (ranks out of step or a peer died)";
        return KNP_E_STATE;
    }
    return KNP_OK;
}

void knp_p2p_free(knp_ctx* ctx) {
    if (!ctx->p2p) return;
    KnpP2P& C = *ctx->p2p;
    for (auto& P : C.plans) {
        for (int r = 0; r < (int)P.peer_box.size(); ++r)
            if (P.peer_box[r] && r != C.rank) (void)hipIpcCloseMemHandle(P.peer_box[r]);
        if (P.box) (void)hipFree(P.box);
        if (P.d_send_idx) (void)hipFree(P.d_send_idx);
        if (P.d_recv_idx) (void)hipFree(P.d_recv_idx);
        if (P.d_rev_dst) (void)hipFree(P.d_rev_dst);
        if (P.d_rev_ptr) (void)hipFree(P.d_rev_ptr);
        if (P.d_rev_pos) (void)hipFree(P.d_rev_pos);
        if (P.d_counter) (void)hipFree(P.d_counter);
    }
    if (C.h_err) (void)hipHostFree(C.h_err);
    if (C.d_err) (void)hipFree(C.d_err);
    delete ctx->p2p;
    ctx->p2p = nullptr;
    ctx->p2p_fine = ctx->p2p_red = -1;
    for (int h = 0; h < KNP_MAX_HIER; ++h)
        for (auto& L : ctx->hier[h].lv) L.p2p_halo = L.p2p_repl = -1;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int knp_p2p_init(knp_ctx* ctx, int32_t rank, int32_t size, double timeout_seconds) {
    if (!ctx) return KNP_E_ARG;
    if (size < 2 || size > KNP_P2P_MAXPEERS || rank < 0 || rank >= size) { ctx->err = "p2p needs 2..16 ranks"; return KNP_E_ARG; }
    knp_p2p_free(ctx);
    ctx->p2p = new KnpP2P();
    KnpP2P& C = *ctx->p2p;
    C.rank = rank; C.size = size;
    if (!(timeout_seconds > 0)) timeout_seconds = 30.0;
    C.timeout_ticks = (int64_t)(timeout_seconds * 1.0e8);   // wall_clock64 runs at 100 MHz
    PCHK(hipHostMalloc((void**)&C.h_err, 64, hipHostMallocMapped));
    *C.h_err = 0;
    PCHK(hipHostGetDevicePointer((void**)&C.h_err_dev, (void*)C.h_err, 0));
    PCHK(hipMalloc((void**)&C.d_err, sizeof(int)));
    PCHK(hipMemset(C.d_err, 0, sizeof(int)));
    return KNP_OK;
}

int knp_p2p_shutdown(knp_ctx* ctx) {
    if (!ctx) return KNP_E_ARG;
    (void)hipDeviceSynchronize();
    knp_p2p_free(ctx);
    return KNP_OK;
}

int knp_p2p_plan_create(knp_ctx* ctx, int32_t kind, int64_t n_fwd, int64_t n_rev, int32_t* plan_out, void* ipc_handle_out) {
    if (!ctx) return KNP_E_ARG;
    if (!ctx->p2p) { ctx->err = "knp_p2p_init first"; return KNP_E_STATE; }
    if ((kind != KNP_P2P_HALO && kind != KNP_P2P_ALLREDUCE) || n_fwd < 0 || n_rev < 0 || !plan_out || !ipc_handle_out) { ctx->err = "bad p2p plan arguments"; return KNP_E_ARG; }
    KnpP2P& C = *ctx->p2p;
    KnpP2PPlan P;
    P.kind = kind; P.n_fwd = n_fwd; P.n_rev = kind == KNP_P2P_HALO ? n_rev : 0;
    P.hdr_bytes = hdr_bytes_of(kind, C.size);
    const size_t n_data = kind == KNP_P2P_HALO ? 2 * (size_t)(n_fwd + n_rev) : 2 * (size_t)C.size * (size_t)n_fwd;
    P.box_bytes = P.hdr_bytes + std::max<size_t>(n_data, 1) * sizeof(double);
    // uncached so that the peers' stores and the local polls / reads never sit in a stale cache line
    if (hipExtMallocWithFlags((void**)&P.box, P.box_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        PCHK(hipExtMallocWithFlags((void**)&P.box, P.box_bytes, hipDeviceMallocFinegrained));
    }
    PCHK(hipMemset(P.box, 0, P.box_bytes));
    PCHK(hipMalloc((void**)&P.d_counter, 2 * sizeof(unsigned int)));
    PCHK(hipMemset(P.d_counter, 0, 2 * sizeof(unsigned int)));
    PCHK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    PCHK(hipIpcGetMemHandle(&h, P.box));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    std::memcpy(ipc_handle_out, &h, 64);
    C.plans.push_back(P);
    *plan_out = (int32_t)C.plans.size() - 1;
    return KNP_OK;
}

int knp_p2p_plan_connect(knp_ctx* ctx, int32_t plan, const void* handles, int32_t n_peers, const int32_t* peer_rank,
                         const int64_t* send_ptr, const int32_t* send_idx, const int64_t* remote_fwd_off, const int64_t* recv_ptr,
                         const int32_t* recv_idx, const int64_t* remote_rev_off) {
    if (!ctx) return KNP_E_ARG;
    if (!ctx->p2p || plan < 0 || plan >= (int)ctx->p2p->plans.size() || !handles) { ctx->err = "bad p2p plan"; return KNP_E_ARG; }
    KnpP2P& C = *ctx->p2p;
    KnpP2PPlan& P = C.plans[plan];
    if (P.connected) { ctx->err = "p2p plan already connected"; return KNP_E_STATE; }
    std::vector<char> need(C.size, 0);
    if (P.kind == KNP_P2P_HALO) {
        if (n_peers < 0 || n_peers > KNP_P2P_MAXPEERS || (n_peers && (!peer_rank || !send_ptr || !recv_ptr || !remote_fwd_off || !remote_rev_off))) { ctx->err = "bad p2p peer list"; return KNP_E_ARG; }
        P.n_peers = n_peers;
        P.send_ptr[0] = P.recv_ptr[0] = 0;
        for (int j = 0; j < n_peers; ++j) {
            const int r = peer_rank[j];
            if (r < 0 || r >= C.size || r == C.rank || (j && r <= peer_rank[j - 1])) { ctx->err = "p2p peers must be ascending, distinct and not this rank"; return KNP_E_ARG; }
            if (send_ptr[j + 1] < send_ptr[j] || recv_ptr[j + 1] < recv_ptr[j] || send_ptr[0] != 0 || recv_ptr[0] != 0) { ctx->err = "p2p pointer arrays not monotone"; return KNP_E_ARG; }
            P.peer_rank[j] = r; need[r] = 1;
            P.send_ptr[j + 1] = send_ptr[j + 1]; P.recv_ptr[j + 1] = recv_ptr[j + 1];
            for (int par = 0; par < 2; ++par) {
                if (remote_fwd_off[2 * j + par] < 0 || remote_rev_off[2 * j + par] < 0) { ctx->err = "negative p2p offset"; return KNP_E_ARG; }
            }
        }
        if ((n_peers ? send_ptr[n_peers] : 0) != P.n_rev || (n_peers ? recv_ptr[n_peers] : 0) != P.n_fwd) { ctx->err = "p2p list lengths differ from the plan"; return KNP_E_ARG; }
        if ((P.n_rev && !send_idx) || (P.n_fwd && !recv_idx)) { ctx->err = "p2p index lists missing"; return KNP_E_ARG; }
        P.remote_fwd_off.assign(remote_fwd_off, remote_fwd_off + 2 * n_peers);
        P.remote_rev_off.assign(remote_rev_off, remote_rev_off + 2 * n_peers);
        PCHK(hipMalloc((void**)&P.d_send_idx, std::max<size_t>(P.n_rev, 1) * sizeof(int32_t)));
        PCHK(hipMalloc((void**)&P.d_recv_idx, std::max<size_t>(P.n_fwd, 1) * sizeof(int32_t)));
        if (P.n_rev) PCHK(hipMemcpy(P.d_send_idx, send_idx, P.n_rev * sizeof(int32_t), hipMemcpyHostToDevice));
        if (P.n_fwd) PCHK(hipMemcpy(P.d_recv_idx, recv_idx, P.n_fwd * sizeof(int32_t), hipMemcpyHostToDevice));
        {   // reverse halo: every owned entry with its mailbox positions in peer order (the send list is grouped by peer)
            std::vector<int32_t> order(P.n_rev);
            for (int64_t k = 0; k < P.n_rev; ++k) order[k] = (int32_t)k;
            std::stable_sort(order.begin(), order.end(), [&](int32_t p, int32_t q) { return send_idx[p] < send_idx[q]; });
            std::vector<int32_t> dst, ptr(1, 0);
            for (int64_t k = 0; k < P.n_rev; ++k) {
                if (k == 0 || send_idx[order[k]] != send_idx[order[k - 1]]) { dst.push_back(send_idx[order[k]]); ptr.push_back(ptr.back()); }
                ptr.back() += 1;
            }
            P.n_rev_dst = (int64_t)dst.size();
            PCHK(hipMalloc((void**)&P.d_rev_dst, std::max<size_t>(dst.size(), 1) * sizeof(int32_t)));
            PCHK(hipMalloc((void**)&P.d_rev_ptr, ptr.size() * sizeof(int32_t)));
            PCHK(hipMalloc((void**)&P.d_rev_pos, std::max<size_t>(order.size(), 1) * sizeof(int32_t)));
            if (!dst.empty()) PCHK(hipMemcpy(P.d_rev_dst, dst.data(), dst.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            PCHK(hipMemcpy(P.d_rev_ptr, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            if (!order.empty()) PCHK(hipMemcpy(P.d_rev_pos, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    } else {
        for (int r = 0; r < C.size; ++r) need[r] = (r != C.rank);
    }
    P.peer_box.assign(C.size, nullptr);
    P.peer_box[C.rank] = P.box;
    for (int r = 0; r < C.size; ++r) {
        if (!need[r]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*)handles + (size_t)r * 64, 64);
        void* ptr = nullptr;
        PCHK(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        P.peer_box[r] = (char*)ptr;
    }
    P.connected = true;
    return KNP_OK;
}

int knp_p2p_attach(knp_ctx* ctx, int32_t what, int32_t hier, int32_t level, int32_t plan) {
    if (!ctx) return KNP_E_ARG;
    if (plan >= 0) {
        const int kind = (what == KNP_P2P_ATTACH_FINE_HALO || what == KNP_P2P_ATTACH_LEVEL_HALO) ? KNP_P2P_HALO : KNP_P2P_ALLREDUCE;
        if (!get_plan(ctx, plan, kind)) return KNP_E_ARG;
    } else {
        plan = -1;
    }
    if (what == KNP_P2P_ATTACH_FINE_HALO) { ctx->p2p_fine = plan; return KNP_OK; }
    if (what == KNP_P2P_ATTACH_SLOTS) {
        if (plan >= 0 && ctx->p2p->plans[plan].n_fwd < 64) { ctx->err = "slot all-reduce plan needs length >= 64"; return KNP_E_ARG; }
        ctx->p2p_red = plan;
        return KNP_OK;
    }
    if (hier < 0 || hier >= KNP_MAX_HIER || level < 0 || level >= ctx->hier[hier].levels) { ctx->err = "bad hierarchy / level"; return KNP_E_ARG; }
    if (what == KNP_P2P_ATTACH_LEVEL_HALO) { ctx->hier[hier].lv[level].p2p_halo = plan; return KNP_OK; }
    if (what == KNP_P2P_ATTACH_LEVEL_REPL) { ctx->hier[hier].lv[level].p2p_repl = plan; return KNP_OK; }
    ctx->err = "unknown p2p attachment";
    return KNP_E_ARG;
}

int knp_p2p_test_halo(knp_ctx* ctx, int32_t plan, double* x, int32_t reverse) {
    if (!ctx || !x) return KNP_E_ARG;
    int rc = reverse ? knp_p2p_halo_reverse(ctx, plan, x) : knp_p2p_halo_forward(ctx, plan, x);
    if (rc != KNP_OK) return rc;
    PCHK(hipStreamSynchronize(ctx->stream));
    return knp_p2p_check(ctx);
}

int knp_p2p_test_allreduce(knp_ctx* ctx, int32_t plan, double* v, int32_t n) {
    if (!ctx || !v) return KNP_E_ARG;
    int rc = knp_p2p_allreduce(ctx, plan, v, n, nullptr, nullptr, 0);
    if (rc != KNP_OK) return rc;
    PCHK(hipStreamSynchronize(ctx->stream));
    return knp_p2p_check(ctx);
}

}  // extern "C"
