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
// one wave polling an epoch word until it reaches `need` (wrap-safe), at most `patience` 10-ns ticks; a timeout is counted, never silent
__global__ void k_band_wait(uint32_t* word, uint32_t need, unsigned long long patience, uint32_t* timeouts) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const uint32_t cur = __hip_atomic_fetch_add(word, 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(cur - need) >= 0) break;
        if (wall_clock64() - t0 > patience) { atomicAdd(timeouts, 1u); break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
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
                       (unsigned long long)timeout_us * 100ull, c->band_sync_own + BAND_TIMEOUT_WORD);
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
                       c->band_sync + BAND_TIMEOUT_WORD);
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

// ---- RCCL transport.  The four entry points it needs are resolved from librccl.so on first use; the communicator is the caller's
// (ncclCommInitRank in the host program).  Types as in <rccl/rccl.h>: ncclResult_t / ncclDataType_t are ints, ncclUint8 == 1.
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_sendrecv_fn)(void*, size_t, int, int, void*, hipStream_t);
static struct { void* lib; nccl_group_fn start, end; nccl_sendrecv_fn send, recv; bool tried; } g_rccl;
static bool rccl_load() {
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
        void* l = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!l) continue;
        g_rccl.start = reinterpret_cast<nccl_group_fn>(dlsym(l, "ncclGroupStart"));
        g_rccl.end = reinterpret_cast<nccl_group_fn>(dlsym(l, "ncclGroupEnd"));
        g_rccl.send = reinterpret_cast<nccl_sendrecv_fn>(dlsym(l, "ncclSend"));
        g_rccl.recv = reinterpret_cast<nccl_sendrecv_fn>(dlsym(l, "ncclRecv"));
        if (g_rccl.start && g_rccl.end && g_rccl.send && g_rccl.recv) { g_rccl.lib = l; return true; }
        dlclose(l);
    }
    return false;
}

int b32_gather_bands_rccl(b32_ctx* c, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1) {
    if (!c || !nccl_comm || !c->fb || !y0 || !y1 || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return B32_E_ARG;
    for (int r = 0; r < nranks; ++r) if (y0[r] > y1[r] || y1[r] > c->height) return B32_E_ARG;
    if (nranks == 1) return B32_OK;
    (void)hipSetDevice(c->device);
    { const int rc = flush_clear(c); if (rc) return rc; }
    if (!rccl_load()) return B32_E_UNSUPPORTED;
    const size_t row = (size_t)c->width * 4;
    constexpr int NCCL_UINT8 = 1;
    unsigned char* base = reinterpret_cast<unsigned char*>(c->fb);
    int e = g_rccl.start();
    if (rank == root) {
        for (int r = 0; r < nranks && !e; ++r)
            if (r != root && y1[r] > y0[r]) e = g_rccl.recv(base + (size_t)y0[r] * row, (size_t)(y1[r] - y0[r]) * row, NCCL_UINT8, r, nccl_comm, c->stream);
    } else if (y1[rank] > y0[rank]) {
        e = g_rccl.send(base + (size_t)y0[rank] * row, (size_t)(y1[rank] - y0[rank]) * row, NCCL_UINT8, root, nccl_comm, c->stream);
    }
    const int e2 = g_rccl.end();
    if (e || e2) { c->last_hip = e ? e : e2; return B32_E_HIP; }
    return B32_OK;
}

}  // extern "C"
