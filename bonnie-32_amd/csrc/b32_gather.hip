// b32_gather.hip -- the C ABI of include/b32raster.h, part 5: the multi-GPU exchange step behind the boundary (BASELINE config C4).
//
// The frame is sharded by screen rows (b32_set_band); the presenter reads fb.pixels of ONE process (game/renderer.rs:179-214), so the
// rows of every band must end up in the root rank's framebuffer.  Two transports, both reachable by a host that knows nothing but
// this C ABI:
//
//   shared framebuffer (b32_band_export / _import / _attach): the root exports its library-owned framebuffer (HIP IPC memory handle, or
//       the plain device pointer inside one process); a band rank BINDS it as its own framebuffer, and its fused fill kernel -- which
//       only ever writes the rows of its band -- stores them straight into the root's HBM over xGMI.  There is no copy and no gather
//       launch at all; what is left of the collective is ordering, carried by one 32-bit epoch word per rank in a second shared block:
//       a rank PUBLISHES frame n behind its kernels (one system-scope store on its stream), the root's stream WAITS for it (a one-wave
//       kernel polling the word), the root RELEASES frame n once it has consumed it and a rank ACQUIRES that before it overwrites its
//       rows with frame n + 1.  Nothing of this touches the host between frames.
//   RCCL (b32_gather_bands_rccl): every rank renders into its own framebuffer and the band rows travel by one grouped
//       ncclSend / ncclRecv per rank on the context's stream (librccl.so is loaded on first use: the library itself has no link
//       dependency on it).  For ranks whose memory cannot be mapped into each other (other nodes, no peer access).
//
// tests/test_gpu_parity.py::test_band_ranks_share_one_gpu[*-ipc] drives the shared-framebuffer transport with real processes (no
// torch in the workers) on ONE GPU; the RCCL transport and peer mappings between two GPUs have not run anywhere yet (no multi-GPU box
// was available to this repository's builds) and say so in DESIGN.md.
#include "b32_host.h"
#include <dlfcn.h>

namespace b32 {

constexpr uint32_t BAND_RANKS = 64;            // epoch words: [rank] published frame, [BAND_RANKS + rank] unused, ROOT_WORD the root's release
constexpr uint32_t BAND_STRIDE = 16;           // words between two epoch words (one 64-byte line each)
constexpr uint32_t BAND_ROOT_WORD = BAND_RANKS * BAND_STRIDE;
constexpr uint32_t BAND_TIMEOUT_WORD = BAND_ROOT_WORD + BAND_STRIDE;
constexpr size_t   BAND_SYNC_BYTES = (BAND_TIMEOUT_WORD + BAND_STRIDE) * sizeof(uint32_t);

// behind everything enqueued before it on the stream: the rank's rows are complete, make them visible system-wide, then say so
__global__ void k_band_publish(uint32_t* word, uint32_t value) {
    // (read-modify-write atomics on both sides: they execute at the memory side, so the reader -- another process, another XCD's L2,
    // possibly another GPU -- can never be served a stale cached line)
    __threadfence_system();
    (void)__hip_atomic_exchange(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one wave polling an epoch word until it reaches `need` (wrap-safe), at most `patience` 10-ns ticks; a timeout is counted in the shared
// tail AND raised in the waiting context's sticky word (bit 4: b32_frame_finish returns B32_E_BAND_TIMEOUT), never silent
__global__ void k_band_wait(uint32_t* word, uint32_t need, unsigned long long patience, uint32_t* timeouts, Ctrl* ctrl) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t cur = __hip_atomic_fetch_add(word, 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(cur - need) >= 0) break;
        if (wall_clock64() - t0 > patience) { atomicAdd(timeouts, 1u); if (ctrl) atomicOr(&ctrl->sticky, 16u); break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
}
// the root's whole side of a frame in ONE launch: lane r (1 <= r < nranks) polls rank r's word; when all have arrived (or given up,
// counted) lane 0 optionally publishes the root's release word.  (Seven one-lane waits + a release are eight launches of ~5 us each on
// the root's stream: more than the exchange itself.)
__global__ void __launch_bounds__(64) k_band_wait_all(uint32_t* sync, uint32_t nranks, uint32_t need, unsigned long long patience, uint32_t release, Ctrl* ctrl) {
    const uint32_t r = threadIdx.x;
    bool done = !(r >= 1 && r < nranks);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (!done) {
            const uint32_t cur = __hip_atomic_fetch_add(sync + (size_t)r * BAND_STRIDE, 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            done = (int32_t)(cur - need) >= 0;
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        if (wall_clock64() - t0 > patience) { if (!done) { atomicAdd(sync + BAND_TIMEOUT_WORD, 1u); if (ctrl) atomicOr(&ctrl->sticky, 16u); } break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
    if (release && r == 0) (void)__hip_atomic_exchange(sync + BAND_ROOT_WORD, need, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace b32

using namespace b32;

static_assert(sizeof(hipIpcMemHandle_t) <= 64, "B32BandShare holds a HIP IPC memory handle in 64 bytes");

// releases whatever b32_band_import / _attach / _export set up (called by b32_band_close and b32_destroy)
void band_close_any(b32_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->band_fb_ipc || c->band_attached) {
        (void)hipStreamSynchronize(c->stream);
        if (c->fb_external && (reinterpret_cast<void*>(c->fb) == c->band_fb_ipc || c->band_attached)) { c->fb_external = false; c->fb = nullptr; c->width = c->height = 0; }
    }
    if (c->band_fb_ipc) (void)hipIpcCloseMemHandle(c->band_fb_ipc);
    c->band_fb_ipc = nullptr; c->band_sync_own = nullptr; c->band_sync = nullptr; c->band_rank = 0; c->band_attached = false;
}

// The epoch words live in the tail of the root's framebuffer allocation (b32_api.hip allocates FB_TAIL_BYTES behind the pixels): one
// allocation, one IPC handle -- a second small hipMalloc of its own could not be opened by another process (hipIpcOpenMemHandle:
// invalid argument; small allocations share a mapping).
static_assert(BAND_SYNC_BYTES <= FB_TAIL_BYTES, "epoch words must fit the framebuffer allocation's tail");
static size_t band_sync_offset(uint32_t w, uint32_t h) { return (((size_t)w * h * 4) + 4095) & ~(size_t)4095; }
static int band_sync_ensure_root(b32_ctx* c) {
    uint32_t* want = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(c->fb_own) + band_sync_offset(c->width, c->height));
    if (c->band_sync_own == want) return B32_OK;
    HIPCHK(c, hipMemset(want, 0, BAND_SYNC_BYTES));
    c->band_sync_own = want; c->band_sync = want; c->band_rank = 0;
    return B32_OK;
}

extern "C" {

int b32_band_export(b32_ctx* c, B32BandShare* out) {
    if (!c || !out || !c->fb_own || c->fb != c->fb_own) return B32_E_ARG;      // the root's framebuffer must be the library's own allocation (b32_fb_new / _resize)
    (void)hipSetDevice(c->device);
    int rc = band_sync_ensure_root(c);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    hipIpcMemHandle_t hm;
    HIPCHK(c, hipIpcGetMemHandle(&hm, c->fb_own));
    memcpy(out->mem, &hm, sizeof(hm));
    out->sync_offset = band_sync_offset(c->width, c->height);
    out->width = c->width; out->height = c->height; out->device = (uint32_t)c->device;
    return B32_OK;
}

int b32_band_import(b32_ctx* c, const B32BandShare* share, uint32_t rank) {
    if (!c || !share || rank == 0 || rank >= BAND_RANKS || !share->width || !share->height) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rc = flush_clear(c); if (rc) return rc; }
    band_close_any(c);
    if (share->sync_offset != band_sync_offset(share->width, share->height)) return B32_E_ARG;
    hipIpcMemHandle_t hm;
    memcpy(&hm, share->mem, sizeof(hm));
    void* pf = nullptr;
    HIPCHK(c, hipIpcOpenMemHandle(&pf, hm, hipIpcMemLazyEnablePeerAccess));
    c->band_fb_ipc = pf; c->band_sync = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(pf) + share->sync_offset); c->band_rank = rank;
    const int rc = b32_fb_bind_device(c, pf, share->width, share->height);
    if (rc) band_close_any(c);
    return rc;
}

int b32_band_attach(b32_ctx* c, b32_ctx* root, uint32_t rank) {
    if (!c || !root || c == root || rank == 0 || rank >= BAND_RANKS || !root->fb_own || root->fb != root->fb_own) return B32_E_ARG;
    (void)hipSetDevice(root->device);
    int rc = band_sync_ensure_root(root);
    if (rc) return rc;
    (void)hipSetDevice(c->device);
    { const int r2 = settle_pending(c); if (r2) return r2; }
    { const int r2 = flush_clear(c); if (r2) return r2; }
    band_close_any(c);
    if (c->device != root->device) {                    // one process, several GPUs: the peer's memory must be reachable from this device
        int can = 0;
        HIPCHK(c, hipDeviceCanAccessPeer(&can, c->device, root->device));
        if (!can) return B32_E_UNSUPPORTED;
        const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { c->last_hip = (int)e; return B32_E_HIP; }
        (void)hipGetLastError();
    }
    c->band_sync = root->band_sync_own; c->band_rank = rank; c->band_attached = true;
    rc = b32_fb_bind_device(c, root->fb_own, root->width, root->height);
    if (rc) band_close_any(c);
    return rc;
}

int b32_band_close(b32_ctx* c) {
    if (!c) return B32_E_ARG;
    band_close_any(c);
    return B32_OK;
}

int b32_band_publish(b32_ctx* c, uint32_t frame_no) {
    if (!c || !c->band_sync || c->band_rank == 0) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = flush_clear(c); if (rc) return rc; }      // (a deferred clear nobody drew over is part of the frame's rows)
    hipLaunchKernelGGL(k_band_publish, dim3(1), dim3(1), 0, c->stream, c->band_sync + (size_t)c->band_rank * BAND_STRIDE, frame_no);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_band_wait(b32_ctx* c, uint32_t rank, uint32_t frame_no, uint32_t timeout_us) {
    if (!c || !c->band_sync_own || rank == 0 || rank >= BAND_RANKS) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    hipLaunchKernelGGL(k_band_wait, dim3(1), dim3(1), 0, c->stream, c->band_sync_own + (size_t)rank * BAND_STRIDE, frame_no,
                       (unsigned long long)timeout_us * 100ull, c->band_sync_own + BAND_TIMEOUT_WORD, c->d_ctrl);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_band_wait_all(b32_ctx* c, uint32_t nranks, uint32_t frame_no, uint32_t timeout_us, int release_after) {
    if (!c || !c->band_sync_own || nranks < 1 || nranks > BAND_RANKS) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    if (release_after) { const int rc = flush_clear(c); if (rc) return rc; }      // (as b32_band_release)
    if (nranks == 1 && !release_after) return B32_OK;
    hipLaunchKernelGGL(k_band_wait_all, dim3(1), dim3(64), 0, c->stream, c->band_sync_own, nranks, frame_no, (unsigned long long)timeout_us * 100ull,
                       release_after ? 1u : 0u, c->d_ctrl);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_band_release(b32_ctx* c, uint32_t frame_no) {
    if (!c || !c->band_sync_own) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = flush_clear(c); if (rc) return rc; }
    hipLaunchKernelGGL(k_band_publish, dim3(1), dim3(1), 0, c->stream, c->band_sync_own + BAND_ROOT_WORD, frame_no);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_band_acquire(b32_ctx* c, uint32_t frame_no, uint32_t timeout_us) {
    if (!c || !c->band_sync || c->band_rank == 0) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    hipLaunchKernelGGL(k_band_wait, dim3(1), dim3(1), 0, c->stream, c->band_sync + BAND_ROOT_WORD, frame_no, (unsigned long long)timeout_us * 100ull,
                       c->band_sync + BAND_TIMEOUT_WORD, c->d_ctrl);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_band_status(b32_ctx* c, uint32_t* epochs, uint32_t* root_epoch, uint32_t* timeouts) {
    if (!c || !c->band_sync) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    std::vector<uint32_t> h(BAND_SYNC_BYTES / sizeof(uint32_t));
    HIPCHK(c, hipMemcpy(h.data(), c->band_sync, BAND_SYNC_BYTES, hipMemcpyDeviceToHost));       // (synchronises with the null stream only)
    if (epochs) for (uint32_t r = 0; r < BAND_RANKS; ++r) epochs[r] = h[(size_t)r * BAND_STRIDE];
    if (root_epoch) *root_epoch = h[BAND_ROOT_WORD];
    if (timeouts) *timeouts = h[BAND_TIMEOUT_WORD];
    return B32_OK;
}

// ---- RCCL transport.  The entry points it needs are resolved from librccl.so on first use; the communicator is the caller's
// (ncclCommInitRank in the host program, or b32_rccl_comm_create below, which makes it with the very library this file loaded).
// Types as in <rccl/rccl.h>: ncclResult_t / ncclDataType_t are ints, ncclUint8 == 1, ncclUniqueId is 128 opaque bytes passed BY VALUE.
struct NcclId { char internal[128]; };
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_sendrecv_fn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_get_id_fn)(NcclId*);
typedef int (*nccl_init_rank_fn)(void**, int, NcclId, int);
typedef int (*nccl_destroy_fn)(void*);
static struct { void* lib; nccl_group_fn start, end; nccl_sendrecv_fn send, recv; nccl_get_id_fn get_id; nccl_init_rank_fn init_rank; nccl_destroy_fn destroy; bool tried; } g_rccl;
static bool rccl_load() {
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    // (a process that already holds an RCCL -- e.g. the copy PyTorch ships -- gets that very copy back for the same soname)
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" }) {
        void* l = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!l) continue;
        g_rccl.start = reinterpret_cast<nccl_group_fn>(dlsym(l, "ncclGroupStart"));
        g_rccl.end = reinterpret_cast<nccl_group_fn>(dlsym(l, "ncclGroupEnd"));
        g_rccl.send = reinterpret_cast<nccl_sendrecv_fn>(dlsym(l, "ncclSend"));
        g_rccl.recv = reinterpret_cast<nccl_sendrecv_fn>(dlsym(l, "ncclRecv"));
        g_rccl.get_id = reinterpret_cast<nccl_get_id_fn>(dlsym(l, "ncclGetUniqueId"));
        g_rccl.init_rank = reinterpret_cast<nccl_init_rank_fn>(dlsym(l, "ncclCommInitRank"));
        g_rccl.destroy = reinterpret_cast<nccl_destroy_fn>(dlsym(l, "ncclCommDestroy"));
        if (g_rccl.start && g_rccl.end && g_rccl.send && g_rccl.recv && g_rccl.get_id && g_rccl.init_rank && g_rccl.destroy) { g_rccl.lib = l; return true; }
        dlclose(l);
    }
    return false;
}

// self_dst_y0 != NO_LOOPBACK: the root ALSO sends its own band to itself and receives it at row self_dst_y0 (the loopback tap)
constexpr uint32_t NO_LOOPBACK = 0xFFFFFFFFu;
static int gather_bands_rccl_any(b32_ctx* c, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1, uint32_t self_dst_y0) {
    if (!c || !nccl_comm || !c->fb || !y0 || !y1 || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return B32_E_ARG;
    for (int r = 0; r < nranks; ++r) if (y0[r] > y1[r] || y1[r] > c->height) return B32_E_ARG;
    const bool loop = self_dst_y0 != NO_LOOPBACK && rank == root && y1[root] > y0[root];
    if (loop) {       // the destination rows must lie inside the framebuffer and must not overlap the rows that are being sent
        const uint32_t n = y1[root] - y0[root];
        if (self_dst_y0 > c->height || n > c->height - self_dst_y0 || (self_dst_y0 < y1[root] && y0[root] < self_dst_y0 + n)) return B32_E_ARG;
    }
    if (nranks == 1 && !loop) return B32_OK;
    (void)hipSetDevice(c->device);
    // (safe mode: a pending frame that may still need a redraw is settled first -- its rows must be the frame's, not an aborted attempt's;
    // deep mode never blocks the host: a dropped frame is reported by b32_frame_finish)
    if (!c->deep_async) { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rc = flush_clear(c); if (rc) return rc; }
    if (!rccl_load()) return B32_E_UNSUPPORTED;
    const size_t row = (size_t)c->width * 4;
    constexpr int NCCL_UINT8 = 1;
    unsigned char* base = reinterpret_cast<unsigned char*>(c->fb);
    int e = g_rccl.start();
    if (rank == root) {
        for (int r = 0; r < nranks && !e; ++r)
            if (r != root && y1[r] > y0[r]) e = g_rccl.recv(base + (size_t)y0[r] * row, (size_t)(y1[r] - y0[r]) * row, NCCL_UINT8, r, nccl_comm, c->stream);
        if (loop && !e) e = g_rccl.send(base + (size_t)y0[root] * row, (size_t)(y1[root] - y0[root]) * row, NCCL_UINT8, root, nccl_comm, c->stream);
        if (loop && !e) e = g_rccl.recv(base + (size_t)self_dst_y0 * row, (size_t)(y1[root] - y0[root]) * row, NCCL_UINT8, root, nccl_comm, c->stream);
    } else if (y1[rank] > y0[rank]) {
        e = g_rccl.send(base + (size_t)y0[rank] * row, (size_t)(y1[rank] - y0[rank]) * row, NCCL_UINT8, root, nccl_comm, c->stream);
    }
    const int e2 = g_rccl.end();
    if (e || e2) { c->last_hip = e ? e : e2; return B32_E_HIP; }
    return B32_OK;
}

int b32_gather_bands_rccl(b32_ctx* c, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1) {
    return gather_bands_rccl_any(c, nccl_comm, rank, nranks, root, y0, y1, NO_LOOPBACK);
}
int b32_gather_bands_rccl_loopback(b32_ctx* c, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1, uint32_t self_dst_y0) {
    if (self_dst_y0 == NO_LOOPBACK) return B32_E_ARG;
    return gather_bands_rccl_any(c, nccl_comm, rank, nranks, root, y0, y1, self_dst_y0);
}

int b32_rccl_unique_id(unsigned char* id128) {
    if (!id128) return B32_E_ARG;
    if (!rccl_load()) return B32_E_UNSUPPORTED;
    NcclId id;
    if (g_rccl.get_id(&id)) return B32_E_HIP;
    memcpy(id128, id.internal, sizeof(id.internal));
    return B32_OK;
}
int b32_rccl_comm_create(b32_ctx* c, const unsigned char* id128, int rank, int nranks, void** comm) {
    if (!c || !id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return B32_E_ARG;
    *comm = nullptr;
    if (!rccl_load()) return B32_E_UNSUPPORTED;
    (void)hipSetDevice(c->device);                   // (the communicator belongs to the context's device)
    NcclId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* out = nullptr;
    const int e = g_rccl.init_rank(&out, nranks, id, rank);
    if (e || !out) { c->last_hip = e; return B32_E_HIP; }
    *comm = out;
    return B32_OK;
}
int b32_rccl_comm_destroy(void* comm) {
    if (!comm) return B32_E_ARG;
    if (!rccl_load()) return B32_E_UNSUPPORTED;
    return g_rccl.destroy(comm) ? B32_E_HIP : B32_OK;
}

}  // extern "C"
