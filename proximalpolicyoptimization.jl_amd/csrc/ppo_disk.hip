// ppo_disk.hip -- out-of-core rollout store (DiskRollouts / DiskDataset: src/rollouts_to_disk.jl:1-171,
// src/dataset.jl:1-82).  Device -> pinned host copies run on a dedicated copy stream, ordered behind the
// producing kernels by events, so the PCIe transfer of step t overlaps the kernels of step t+1; a writer
// thread drains the pinned ring into one append-only file.
//
// File format (little endian): header {magic "PPOR", u32 version, i64 N, i32 H, i32 F, i32 A, i32 V, i64 T}
// (version 2: then {u64 FNV-1a of the template-vertex table, i32 env kind, i32 0}, checked on load)
// then T step records  [states][active N u32][actions N i32][p_sel N f32][rewards N f32][done N u8]
// then the returns column [T][N] f32 (written after compute_returns).
//   version 1: states = the expanded observations, N*H*F int8 (2304 B per env-step for Q = 8)
//   version 2: states = the env snapshots they are derived from, N*2V int8 (score[V] then degree[V]: 64 B for Q = 8) --
//              the default while streaming (ppo_set_rollout_compact): 81 instead of 2321 bytes per env-step over PCIe
#include "ppo_internal.h"
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <sys/stat.h>
#include <sys/uio.h>
#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdlib>
#include <cerrno>

struct DiskHeader { char magic[4]; uint32_t version; int64_t N; int32_t H, F, A, V; int64_t T; };
// version 2 only, directly behind the header: FNV-1a of the [H][36] template-vertex table the snapshots are expanded with.
// A snapshot file stores vertex scores / degrees, not observations: loading it into a buffer created for another env
// kind (another table) would silently re-derive different rows.
struct DiskTemplateTag { uint64_t tmpl_hash; int32_t env_kind; int32_t reserved; };
static int32_t template_hash(const ppo_rollouts_s* ro, uint64_t* out) {
    std::vector<int8_t> t((size_t)ro->H * PPO_TPL);
    HIP_TRY(hipMemcpyAsync(t.data(), ro->tmpl.p, t.size(), hipMemcpyDeviceToHost, ppo_stream()));
    HIP_TRY(hipStreamSynchronize(ppo_stream()));
    uint64_t h = 1469598103934665603ull;
    for (int8_t b : t) { h ^= (uint8_t)b; h *= 1099511628211ull; }
    *out = h;
    return PPO_OK;
}

struct DiskSink {
    std::string dir;
    int slots = 0, slots0 = 0;            // ring size now / as attached (the deferred-finish mode grows it per collection)
    size_t rec_bytes = 0;
    std::vector<char*> pinned;            // ring of pinned host records
    std::vector<hipEvent_t> produced;     // recorded on the compute stream after step t's kernels
    std::vector<hipEvent_t> copied;       // recorded on the copy stream after the D2H of the slot
    hipStream_t copy_stream = nullptr;
    FILE* f = nullptr;
    std::thread writer;
    std::mutex mu;
    std::condition_variable cv;
    int64_t enq = 0, written = 0;         // records handed to the copy stream / written to the file
    int batch = 1;                        // records per writev()
    bool stop = false;
    std::atomic<bool> failed{false};      // set by the writer thread outside the lock, read by the host thread
    // deferred finish (ppo_set_disk_async): the ring holds the whole collection, ppo_collect_rollouts returns as soon as the
    // last step is on the copy stream, and the writer thread appends the returns column and closes the file by itself
    bool async_fin = false;               // this collection finishes in the writer thread
    bool fin_pending = false, fin_done = false;
    int64_t T_total = 0;
    char* ret_pinned = nullptr; size_t ret_bytes = 0;
    hipEvent_t ret_ready = nullptr, ret_copied = nullptr;
};

// ppo_set_disk_async / PPO_DISK_ASYNC: 1 = a streamed collection returns without waiting for the file (ppo_rollouts_disk_sync,
// the next collection, detach and destroy wait for it); the pinned ring then holds every step of the collection when
// that fits PPO_DISK_ASYNC_MAX_BYTES (default 1 GiB), so the disk never holds the collection back
static int g_disk_async = [] { const char* v = getenv("PPO_DISK_ASYNC"); return v ? atoi(v) != 0 : 0; }();
extern "C" int32_t ppo_set_disk_async(int32_t mode) { g_disk_async = mode < 0 ? 0 : (mode != 0); return PPO_OK; }
static size_t disk_async_budget() { const char* v = getenv("PPO_DISK_ASYNC_MAX_BYTES"); return v ? (size_t)atoll(v) : ((size_t)1 << 30); }

// Process-wide pool of pinned records: the reference creates a fresh DiskRollouts every PPO iteration
// (src/train.jl:185), and hipHostMalloc / hipHostFree of the ring cost more than streaming a short rollout.
static std::mutex g_pool_mu;
static std::vector<std::pair<size_t, char*>> g_pinned_pool;
static char* pinned_get(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pinned_pool.size(); ++i)
            if (g_pinned_pool[i].first == bytes) { char* p = g_pinned_pool[i].second; g_pinned_pool.erase(g_pinned_pool.begin() + i); return p; }
    }
    char* p = nullptr;
    return hipHostMalloc((void**)&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
static void pinned_put(size_t bytes, char* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t held = 0;
    for (auto& e : g_pinned_pool) held += e.first;
    if (held + bytes > ((size_t)2 << 30)) { (void)hipHostFree(p); return; }      // keep at most 2 GiB pinned
    g_pinned_pool.emplace_back(bytes, p);
}

static int rm_rf(const std::string& path) {
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) return 0;
    if (S_ISDIR(st.st_mode)) {
        DIR* d = opendir(path.c_str());
        if (!d) return -1;
        while (dirent* e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            if (rm_rf(path + "/" + e->d_name) != 0) { closedir(d); return -1; }
        }
        closedir(d);
        return rmdir(path.c_str());
    }
    return unlink(path.c_str());
}

// One writer thread, one append-only file.  It takes EVERY record that is ready in one writev() (up to half the ring):
// the file system's per-call work (inode lock, size extension, extent bookkeeping) is paid once per batch instead of once
// per 5 MB record -- a 152 MB expanded record went out at 9.8 GB/s where the 5.3 MB compact ones managed 4.4 GB/s.
static bool write_all(int fd, struct iovec* iov, int n) {
    while (n > 0) {
        ssize_t w = writev(fd, iov, n);
        if (w < 0 && errno == EINTR) continue;          // interrupted before anything was written: retry
        if (w <= 0) return false;                       // error, or a device that accepts nothing (would spin forever)
        while (n > 0 && (size_t)w >= iov->iov_len) { w -= (ssize_t)iov->iov_len; ++iov; --n; }
        if (n > 0 && w > 0) { iov->iov_base = (char*)iov->iov_base + w; iov->iov_len -= (size_t)w; }
    }
    return true;
}
static void writer_loop(DiskSink* s) {
    const int fd = fileno(s->f);
    const int maxb = s->batch > 0 ? s->batch : 1;
    std::vector<struct iovec> iov((size_t)maxb);
    for (;;) {
        int64_t k, avail;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop || s->written < s->enq || (s->fin_pending && !s->fin_done); });
            if (s->written >= s->enq) {
                if (s->fin_pending && !s->fin_done && s->written >= s->T_total) {
                    // deferred finish: the returns column (copied device -> pinned behind the return scan), then close
                    lk.unlock();
                    bool ok = !s->failed && hipEventSynchronize(s->ret_copied) == hipSuccess;
                    if (ok) { struct iovec v; v.iov_base = s->ret_pinned; v.iov_len = s->ret_bytes; ok = write_all(fd, &v, 1); }
                    lk.lock();
                    if (!ok) s->failed = true;
                    s->fin_done = true;
                    lk.unlock();
                    s->cv.notify_all();
                    continue;
                }
                if (s->stop) return;
                continue;
            }
            k = s->written; avail = s->enq - s->written;
        }
        const int nb = (int)std::min<int64_t>(avail, maxb);
        for (int i = 0; i < nb; ++i) {
            const int slot = (int)((k + i) % s->slots);
            if (hipEventSynchronize(s->copied[slot]) != hipSuccess) s->failed = true;     // D2H of this record landed
            iov[(size_t)i].iov_base = s->pinned[slot]; iov[(size_t)i].iov_len = s->rec_bytes;
        }
        if (!s->failed && !write_all(fd, iov.data(), nb)) s->failed = true;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->written = k + nb;
        }
        s->cv.notify_all();
    }
}

// a deferred finish is complete (or has failed) when this returns; the writer thread is stopped and the file closed
static bool disk_sink_settle(DiskSink* s) {
    bool ok = true;
    if (s->writer.joinable()) {
        {
            std::unique_lock<std::mutex> lk(s->mu);
            if (s->fin_pending) s->cv.wait(lk, [&] { return s->fin_done || s->failed; });
            s->stop = true;
        }
        s->cv.notify_all();
        s->writer.join();
    }
    if (s->fin_pending) { ok = !s->failed && s->fin_done; s->fin_pending = false; }
    if (s->f) { if (fclose(s->f) != 0) ok = false; s->f = nullptr; }
    return ok;
}

void disk_sink_destroy(DiskSink* s) {
    if (!s) return;
    (void)disk_sink_settle(s);
    if (s->f) fclose(s->f);
    // a device -> host copy may still be in flight into a record (error / early-detach paths): the records go back to a
    // process-wide pool, so nobody else may receive one before the copy stream has drained
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    for (char* p : s->pinned) pinned_put(s->rec_bytes, p);
    pinned_put(s->ret_bytes, s->ret_pinned);
    if (s->ret_ready) (void)hipEventDestroy(s->ret_ready);
    if (s->ret_copied) (void)hipEventDestroy(s->ret_copied);
    for (auto e : s->produced) (void)hipEventDestroy(e);
    for (auto e : s->copied) (void)hipEventDestroy(e);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    delete s;
}

ppo_rollouts_s::~ppo_rollouts_s() { disk_sink_destroy(sink); sink = nullptr; }

static size_t state_bytes(const ppo_rollouts_s* ro, bool compact) {
    return (size_t)ro->N * (compact ? (size_t)2 * ro->V : (size_t)ro->H * ro->F);
}
static size_t record_bytes(const ppo_rollouts_s* ro, bool compact) {
    const size_t N = (size_t)ro->N;
    return state_bytes(ro, compact) + N * 4 * 4 + N;
}

extern "C" int32_t ppo_rollouts_attach_disk(ppo_rollouts_t ro, const char* dir, int32_t pinned_slots) {
    ARG_CHECK(ro && dir && dir[0], "DiskRollouts: bad argument");
    ARG_CHECK(pinned_slots >= 2 && pinned_slots <= 64, "DiskRollouts: 2..64 pinned slots");
    if (ro->sink) { disk_sink_destroy(ro->sink); ro->sink = nullptr; }
    // prepare_state_data_directory: wipe and recreate (src/rollouts_to_disk.jl:7-13)
    if (rm_rf(dir) != 0) { ppo_set_error(std::string("DiskRollouts: cannot clear ") + dir); return PPO_ERR_ARG; }
    if (mkdir(dir, 0777) != 0) { ppo_set_error(std::string("DiskRollouts: cannot create ") + dir); return PPO_ERR_ARG; }
    const std::string states = std::string(dir) + "/states";
    if (mkdir(states.c_str(), 0777) != 0) { ppo_set_error("DiskRollouts: cannot create states/"); return PPO_ERR_ARG; }
    DiskSink* s = new DiskSink();
    s->dir = dir; s->slots = pinned_slots; s->rec_bytes = 0;           // the pinned ring is sized in disk_sink_begin
    s->pinned.assign(pinned_slots, nullptr);
    s->produced.resize(pinned_slots); s->copied.resize(pinned_slots);
    for (int i = 0; i < pinned_slots; ++i) {
        (void)hipEventCreateWithFlags(&s->produced[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&s->copied[i], hipEventDisableTiming);
    }
    if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        disk_sink_destroy(s); ppo_set_error("DiskRollouts: copy stream"); return PPO_ERR_HIP;
    }
    ro->sink = s;
    return PPO_OK;
}

extern "C" int32_t ppo_rollouts_detach_disk(ppo_rollouts_t ro) {
    ARG_CHECK(ro, "null");
    disk_sink_destroy(ro->sink);
    ro->sink = nullptr;
    return PPO_OK;
}

int disk_sink_slots(const ppo_rollouts_s* ro) { return ro->sink ? ro->sink->slots : 0; }
// steps per persistent launch of a streamed collection: half the ring in flight, at most 8 (finer D2H / launch overlap)
int disk_sink_chunk(const ppo_rollouts_s* ro) { return ro->sink ? std::max(1, std::min(ro->sink->slots / 2, 8)) : 0; }

extern "C" int32_t ppo_rollouts_disk_sync(ppo_rollouts_t ro) {
    ARG_CHECK(ro, "null");
    if (!ro->sink) return PPO_OK;
    if (!disk_sink_settle(ro->sink)) { ppo_set_error("DiskRollouts: writer failed (disk full?)"); return PPO_ERR_ARG; }
    return PPO_OK;
}

int32_t disk_sink_begin(ppo_rollouts_s* ro, int64_t T) {
    DiskSink* s = ro->sink;
    if (!s) return PPO_OK;
    (void)disk_sink_settle(s);                        // a previous collection (its deferred finish included): restart the file
    s->enq = s->written = 0; s->stop = false; s->failed = false; s->fin_pending = s->fin_done = false;
    const size_t rec = record_bytes(ro, ro->compact);           // the storage form is decided per collection
    if (s->copy_stream) HIP_TRY(hipStreamSynchronize(s->copy_stream));   // nothing of a previous collection is still landing
    s->async_fin = g_disk_async != 0;
    s->T_total = T;
    int want_slots = s->slots0 ? s->slots0 : s->slots;
    if (!s->slots0) s->slots0 = s->slots;
    if (s->async_fin) {                                         // the whole collection in the ring when the budget allows
        const int64_t fit = (int64_t)(disk_async_budget() / (rec ? rec : 1));
        want_slots = (int)std::max<int64_t>(s->slots0, std::min<int64_t>(T, std::max<int64_t>(fit, 2)));
    }
    if (want_slots != s->slots) {
        for (char*& p : s->pinned) { pinned_put(s->rec_bytes, p); p = nullptr; }
        s->rec_bytes = 0;
        for (auto e : s->produced) (void)hipEventDestroy(e);
        for (auto e : s->copied) (void)hipEventDestroy(e);
        s->slots = want_slots;
        s->pinned.assign(want_slots, nullptr);
        s->produced.assign(want_slots, nullptr); s->copied.assign(want_slots, nullptr);
        for (int i = 0; i < want_slots; ++i) {
            (void)hipEventCreateWithFlags(&s->produced[i], hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&s->copied[i], hipEventDisableTiming);
        }
    }
    if (rec != s->rec_bytes) {
        for (char*& p : s->pinned) { pinned_put(s->rec_bytes, p); p = nullptr; }
        s->rec_bytes = rec;
        for (int i = 0; i < s->slots; ++i)
            if (!(s->pinned[i] = pinned_get(rec))) {
                // hand back what was obtained and forget the size, so the next begin allocates again instead of
                // finding rec == rec_bytes with null slots
                for (char*& p : s->pinned) { pinned_put(rec, p); p = nullptr; }
                s->rec_bytes = 0;
                ppo_set_error("DiskRollouts: pinned allocation failed");
                return PPO_ERR_HIP;
            }
    }
    const std::string path = s->dir + "/rollout.bin";
    s->f = fopen(path.c_str(), "wb");
    if (!s->f) { ppo_set_error("DiskRollouts: cannot open " + path); return PPO_ERR_ARG; }
    DiskHeader h;
    memcpy(h.magic, "PPOR", 4); h.version = ro->compact ? 2 : 1; h.N = ro->N; h.H = ro->H; h.F = ro->F; h.A = ro->A; h.V = ro->V; h.T = T;
    bool hdr_ok = fwrite(&h, sizeof(h), 1, s->f) == 1;
    size_t hdr_bytes = sizeof(h);
    if (ro->compact) {
        DiskTemplateTag tag = {0, 0, 0};
        PPO_TRY(template_hash(ro, &tag.tmpl_hash));
        hdr_ok = hdr_ok && fwrite(&tag, sizeof(tag), 1, s->f) == 1;
        hdr_bytes += sizeof(tag);
    }
    if (!hdr_ok || fflush(s->f) != 0) { ppo_set_error("DiskRollouts: header write failed"); return PPO_ERR_ARG; }
    // records go out through the descriptor (writev), the returns column and the close through the FILE again: the stream is
    // flushed here and holds nothing in between.  Blocks reserved up front where the file system can (no per-call
    // allocation; KEEP_SIZE: the file still grows by appending, a short collection leaves no zero tail)
    (void)fallocate(fileno(s->f), FALLOC_FL_KEEP_SIZE, 0, (off_t)(hdr_bytes + (size_t)T * rec + (size_t)T * ro->N * 4));
    s->batch = std::max(1, std::min(s->slots / 2, 8));
    if (const char* e = getenv("PPO_DISK_BATCH")) s->batch = std::max(1, std::min(s->slots, atoi(e)));
    s->writer = std::thread(writer_loop, s);
    return PPO_OK;
}

int32_t disk_sink_step(ppo_rollouts_s* ro, int64_t t) {
    DiskSink* s = ro->sink;
    if (!s) return PPO_OK;
    const int slot = (int)(t % s->slots);
    {   // the slot must have been written out before it is overwritten (back-pressure from the disk)
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->failed || s->written + s->slots > t; });
    }
    if (s->failed) { ppo_set_error("DiskRollouts: writer failed (disk full?)"); return PPO_ERR_ARG; }
    const size_t N = (size_t)ro->N, sb = state_bytes(ro, ro->compact);
    HIP_TRY(hipEventRecord(s->produced[slot], ppo_stream()));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->produced[slot], 0));
    char* dst = s->pinned[slot];
    HIP_TRY(hipMemcpyAsync(dst, (ro->compact ? ro->cstate.p : ro->states.p) + (size_t)t * sb, sb, hipMemcpyDeviceToHost, s->copy_stream)); dst += sb;
    HIP_TRY(hipMemcpyAsync(dst, ro->active.p + t * N, N * 4, hipMemcpyDeviceToHost, s->copy_stream)); dst += N * 4;
    HIP_TRY(hipMemcpyAsync(dst, ro->actions.p + t * N, N * 4, hipMemcpyDeviceToHost, s->copy_stream)); dst += N * 4;
    HIP_TRY(hipMemcpyAsync(dst, ro->p_sel.p + t * N, N * 4, hipMemcpyDeviceToHost, s->copy_stream)); dst += N * 4;
    HIP_TRY(hipMemcpyAsync(dst, ro->rewards.p + t * N, N * 4, hipMemcpyDeviceToHost, s->copy_stream)); dst += N * 4;
    HIP_TRY(hipMemcpyAsync(dst, ro->done.p + t * N, N, hipMemcpyDeviceToHost, s->copy_stream));
    HIP_TRY(hipEventRecord(s->copied[slot], s->copy_stream));
    { std::lock_guard<std::mutex> lk(s->mu); s->enq = t + 1; }
    s->cv.notify_all();
    return PPO_OK;
}

int32_t disk_sink_finish(ppo_rollouts_s* ro) {
    DiskSink* s = ro->sink;
    if (!s) return PPO_OK;
    if (s->async_fin) {
        // deferred finish: the returns column goes device -> pinned on the copy stream behind the return scan; the writer
        // thread appends it after the last record and closes the file (ppo_rollouts_disk_sync waits for that)
        const size_t nb = (size_t)ro->T * ro->N * 4;
        if (nb != s->ret_bytes) {
            pinned_put(s->ret_bytes, s->ret_pinned);
            s->ret_pinned = pinned_get(nb); s->ret_bytes = s->ret_pinned ? nb : 0;
            if (!s->ret_pinned) { ppo_set_error("DiskRollouts: pinned allocation failed (returns column)"); return PPO_ERR_HIP; }
        }
        if (!s->ret_ready) { HIP_TRY(hipEventCreateWithFlags(&s->ret_ready, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->ret_copied, hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(s->ret_ready, ppo_stream()));
        HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ret_ready, 0));
        HIP_TRY(hipMemcpyAsync(s->ret_pinned, ro->returns.p, nb, hipMemcpyDeviceToHost, s->copy_stream));
        HIP_TRY(hipEventRecord(s->ret_copied, s->copy_stream));
        // the finish is due once every record handed over so far is in the file (== ro->T steps unless a step failed to
        // enqueue: the count the writer can actually reach, so a short collection can never leave it waiting)
        { std::lock_guard<std::mutex> lk(s->mu); s->T_total = s->enq; s->fin_pending = true; }
        s->cv.notify_all();
        return PPO_OK;
    }
    {   // drain the ring
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->failed || s->written >= s->enq; });
        s->stop = true;
    }
    s->cv.notify_all();
    s->writer.join();
    if (s->failed) { ppo_set_error("DiskRollouts: writer failed"); return PPO_ERR_ARG; }
    // returns column (the reference rewrites trajectory.csv with the returns at this point)
    const size_t n = (size_t)ro->T * ro->N;
    std::vector<float> ret(n);
    HIP_TRY(hipMemcpyAsync(ret.data(), ro->returns.p, n * 4, hipMemcpyDeviceToHost, ppo_stream()));
    HIP_TRY(hipStreamSynchronize(ppo_stream()));
    if (fseek(s->f, 0, SEEK_END) != 0 || fwrite(ret.data(), 4, n, s->f) != n) { ppo_set_error("DiskRollouts: returns write failed"); return PPO_ERR_ARG; }
    fclose(s->f); s->f = nullptr;
    return PPO_OK;
}

extern "C" int32_t ppo_rollouts_load_disk(ppo_rollouts_t ro, const char* dir) {
    ARG_CHECK(ro && dir, "DiskDataset: bad argument");
    const std::string path = std::string(dir) + "/rollout.bin";
    FILE* f = fopen(path.c_str(), "rb");
    ARG_CHECK(f != nullptr, "DiskDataset: trajectory file missing (src/dataset.jl:7)");
    DiskHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "PPOR", 4) != 0 || (h.version != 1 && h.version != 2)) {
        fclose(f); ppo_set_error("DiskDataset: bad header"); return PPO_ERR_ARG;
    }
    const bool compact = h.version == 2;
    if (h.N != ro->N || h.H != ro->H || h.F != ro->F || (compact && h.V != ro->V)) { fclose(f); ppo_set_error("DiskDataset: shape mismatch"); return PPO_ERR_ARG; }
    if (compact) {
        DiskTemplateTag tag;
        uint64_t mine = 0;
        if (fread(&tag, sizeof(tag), 1, f) != 1) { fclose(f); ppo_set_error("DiskDataset: bad header"); return PPO_ERR_ARG; }
        const int32_t hs = template_hash(ro, &mine);
        if (hs != PPO_OK) { fclose(f); return hs; }
        if (tag.tmpl_hash != mine) {
            fclose(f);
            ppo_set_error("DiskDataset: the env snapshots were written for another env (template table differs from this buffer's)");
            return PPO_ERR_ARG;
        }
    }
    const int64_t T = h.T;
    const size_t N = (size_t)ro->N, sb = state_bytes(ro, compact), rec = record_bytes(ro, compact);
    std::vector<char> buf(rec);
    std::vector<int8_t> st((size_t)T * sb);
    std::vector<uint32_t> act((size_t)T * N);
    std::vector<int32_t> a0((size_t)T * N);
    std::vector<float> ps((size_t)T * N), rw((size_t)T * N), ret((size_t)T * N);
    std::vector<uint8_t> dn((size_t)T * N);
    for (int64_t t = 0; t < T; ++t) {
        if (fread(buf.data(), 1, rec, f) != rec) { fclose(f); ppo_set_error("DiskDataset: truncated file"); return PPO_ERR_ARG; }
        const char* p = buf.data();
        memcpy(st.data() + (size_t)t * sb, p, sb); p += sb;
        memcpy(act.data() + t * N, p, N * 4); p += N * 4;
        memcpy(a0.data() + t * N, p, N * 4); p += N * 4;
        memcpy(ps.data() + t * N, p, N * 4); p += N * 4;
        memcpy(rw.data() + t * N, p, N * 4); p += N * 4;
        memcpy(dn.data() + t * N, p, N);
    }
    const bool have_ret = fread(ret.data(), 4, (size_t)T * N, f) == (size_t)T * N;
    fclose(f);
    ARG_CHECK(have_ret, "DiskDataset: returns column missing (collection did not finish)");
    if (!compact) {
        PPO_TRY(ppo_rollouts_set(ro, T, st.data(), act.data(), a0.data(), ps.data(), ret.data(), dn.data()));
    } else {                                      // env snapshots go back as they are: the buffer stays in the compact form
        PPO_TRY(rollouts_reserve(ro, T, true));
        ro->T = T; ro->adv_T = -1;
        PPO_TRY(set_index_all(ro));                  // before the copies below: no early return while they are in flight
        const size_t n = (size_t)T * N;
        hipStream_t s = ppo_stream();
        // the sources are local vectors: whatever happens, the stream is drained before this function returns
        hipError_t e = hipMemcpyAsync(ro->cstate.p, st.data(), st.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ro->active.p, act.data(), n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ro->actions.p, a0.data(), n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ro->p_sel.p, ps.data(), n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ro->returns.p, ret.data(), n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ro->done.p, dn.data(), n, hipMemcpyHostToDevice, s);
        const hipError_t es = hipStreamSynchronize(s);
        HIP_TRY(e);
        HIP_TRY(es);
    }
    const hipError_t e2 = hipMemcpyAsync(ro->rewards.p, rw.data(), (size_t)T * N * 4, hipMemcpyHostToDevice, ppo_stream());
    const hipError_t es2 = hipStreamSynchronize(ppo_stream());
    HIP_TRY(e2);
    HIP_TRY(es2);
    return PPO_OK;
}
