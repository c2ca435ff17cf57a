// Collectives of the sharded run: RCCL (opened at run time) and the in-process transport.
// See comm.h.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only; the library is opened with dlopen
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "context.h"        // DevBuf / PinnedBuf

namespace rala_hip {

namespace {

// ---------------------------------------------------------------------------------------------
// RCCL
// ---------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        // a copy that is already in the process (PyTorch brings its own) wins: one RCCL per process
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.lib) break;
        }
        for (size_t k = 0; !api.lib && k < sizeof(names) / sizeof(names[0]); ++k) api.lib = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) {
            api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(api.lib, name);
            if (!p) { ok = false; api.err = std::string("librccl lacks ") + name; }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    });
    return &api;
}

class RcclComm;

// the communicators of THIS process by group id: a rank that fails between two collectives aborts
// the ones of its group that live here (ranks as threads), so that nobody waits for it for ever
struct RcclRegistry {
    std::mutex m;
    std::vector<RcclComm*> comms;
};
RcclRegistry& rccl_registry() {
    static RcclRegistry r;
    return r;
}

class RcclComm : public Comm {
public:
    RcclComm(uint32_t rank, uint32_t world) { rank_ = rank; world_ = world; }
    ~RcclComm() override {
        {
            RcclRegistry& reg = rccl_registry();
            std::lock_guard<std::mutex> hold(reg.m);
            for (size_t k = 0; k < reg.comms.size(); ++k) {
                if (reg.comms[k] == this) { reg.comms.erase(reg.comms.begin() + k); break; }
            }
        }
        // (a peer's abort() has taken the handle away already: ncclCommAbort frees it)
        const ncclComm_t mine = comm_.exchange(nullptr);
        if (mine) (void)(aborted_ ? api_->CommAbort(mine) : api_->CommDestroy(mine));
    }
    bool init(const void* id, std::string* err) {
        api_ = rccl_api();
        if (!api_->lib) { *err = api_->err; return false; }
        static_assert(sizeof(ncclUniqueId) == kCommIdBytes, "unique id size");
        memcpy(&uid_, id, sizeof(uid_));
        ncclComm_t made = nullptr;
        const ncclResult_t r = api_->CommInitRank(&made, (int)world_, uid_, (int)rank_);
        if (r != ncclSuccess) { *err = std::string("ncclCommInitRank: ") + api_->GetErrorString(r); return false; }
        comm_ = made;
        RcclRegistry& reg = rccl_registry();
        std::lock_guard<std::mutex> hold(reg.m);
        reg.comms.push_back(this);
        return true;
    }

    // Called by a rank that cannot go on.  Its peers inside this process are released from whatever
    // collective they wait in (their calls fail from here on); peers in other processes are the
    // launcher's to end when this process exits with an error.
    // A peer's handle is taken away (exchanged for null) before it is aborted - ncclCommAbort frees it, and its owner
    // must neither use nor free it again.  The owner enqueues under call_m_: the abort waits for an enqueue that is under
    // way, and no call starts on a handle that is gone; an owner that does not come out of its call (the first
    // collective connects inside ncclGroupEnd and waits for the rank that failed) is aborted all the same after a
    // moment - releasing a blocked call from another thread is what ncclCommAbort is for.
    void abort() override {
        RcclRegistry& reg = rccl_registry();
        std::lock_guard<std::mutex> hold(reg.m);
        for (RcclComm* c : reg.comms) {
            if (memcmp(&c->uid_, &uid_, sizeof(uid_)) != 0 || c->aborted_.exchange(true)) continue;
            if (c == this) continue;
            const bool quiet = c->call_m_.try_lock_for(std::chrono::milliseconds(200));
            const ncclComm_t theirs = c->comm_.exchange(nullptr);
            if (theirs) (void)api_->CommAbort(theirs);
            if (quiet) c->call_m_.unlock();
        }
    }

    int host_all_gather(const uint64_t* mine, uint32_t n, uint64_t* all, hipStream_t s) override {
        const size_t total = (size_t)world_ * n;
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        if (d_small_.ensure(total + n) != hipSuccess || p_small_.ensure(total + n) != hipSuccess) return fail("out of memory");
        uint64_t* h = p_small_.p;                        // [0, n): mine; [n, n + total): all
        memcpy(h, mine, (size_t)n * 8);
        if (hipMemcpyAsync(d_small_.p, h, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess) return fail("copy");
        if (!ok(api_->AllGather(d_small_.p, d_small_.p + n, n, ncclUint64, comm, s), "ncclAllGather")) return -1;
        call.unlock();
        if (hipMemcpyAsync(h + n, d_small_.p + n, total * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return fail("copy");
        if (hipStreamSynchronize(s) != hipSuccess) return fail("hipStreamSynchronize");
        memcpy(all, h + n, total * 8);
        return 0;
    }

    int all_to_all_v(const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                     size_t elem_bytes, hipStream_t s, bool own_part) override {
        size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        if (!ok(api_->GroupStart(), "ncclGroupStart")) return -1;
        bool good = true;                               // (a group that was opened is always closed)
        for (uint32_t p = 0; p < world_ && good; ++p) {
            const size_t sb = (size_t)send_counts[p] * elem_bytes;
            const size_t rb = p == rank_ && !own_part ? 0 : (size_t)recv_counts[p] * elem_bytes;
            if (p == rank_) {
                self_so = so; self_ro = ro;
            } else {
                if (sb) good = ok(api_->Send((const char*)send + so, sb, ncclUint8, (int)p, comm, s), "ncclSend");
                if (rb && good) good = ok(api_->Recv((char*)recv + ro, rb, ncclUint8, (int)p, comm, s), "ncclRecv");
            }
            so += sb; ro += rb;
        }
        if (!close_group(good)) return -1;
        const size_t mine = own_part ? (size_t)send_counts[rank_] * elem_bytes : 0;
        if (mine && hipMemcpyAsync((char*)recv + self_ro, (const char*)send + self_so, mine, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            return fail("copy of the own part");
        }
        return 0;
    }

    int all_gather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
        if (bytes == 0) return 0;
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        return ok(api_->AllGather(send, recv, bytes, ncclUint8, comm, s), "ncclAllGather") ? 0 : -1;
    }

    int all_gather_v(const void* send, void* recv, const uint64_t* counts, size_t elem_bytes, hipStream_t s) override {
        const size_t mine = (size_t)counts[rank_] * elem_bytes;
        size_t ro = 0, self_ro = 0;
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        if (!ok(api_->GroupStart(), "ncclGroupStart")) return -1;
        bool good = true;
        for (uint32_t p = 0; p < world_ && good; ++p) {
            const size_t rb = (size_t)counts[p] * elem_bytes;
            if (p == rank_) {
                self_ro = ro;
            } else {
                if (mine) good = ok(api_->Send(send, mine, ncclUint8, (int)p, comm, s), "ncclSend");
                if (rb && good) good = ok(api_->Recv((char*)recv + ro, rb, ncclUint8, (int)p, comm, s), "ncclRecv");
            }
            ro += rb;
        }
        if (!close_group(good)) return -1;
        if (mine && (const char*)send != (char*)recv + self_ro &&
            hipMemcpyAsync((char*)recv + self_ro, send, mine, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            return fail("copy of the own part");
        }
        return 0;
    }

    int all_reduce_u32(uint32_t* buf, size_t n, ReduceOp op, hipStream_t s) override {
        if (n == 0) return 0;
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        const ncclRedOp_t r = op == ReduceOp::kSum ? ncclSum : op == ReduceOp::kMin ? ncclMin : ncclMax;
        return ok(api_->AllReduce(buf, buf, n, ncclUint32, r, comm, s), "ncclAllReduce") ? 0 : -1;
    }

    int barrier(hipStream_t s) override {
        std::unique_lock<std::timed_mutex> call(call_m_);
        const ncclComm_t comm = live();
        if (!comm) return -1;
        if (d_small_.ensure(8) != hipSuccess) return fail("out of memory");
        if (!ok(api_->AllReduce(d_small_.p, d_small_.p, 1, ncclUint64, ncclSum, comm, s), "ncclAllReduce")) return -1;
        call.unlock();
        return hipStreamSynchronize(s) == hipSuccess ? 0 : fail("hipStreamSynchronize");
    }

private:
    bool ok(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return true;
        err_ = std::string(what) + ": " + api_->GetErrorString(r);
        return false;
    }
    int fail(const char* what) { err_ = what; return -1; }
    // the handle, or null when the group was aborted (called with call_m_ held)
    ncclComm_t live() {
        const ncclComm_t c = aborted_.load() ? nullptr : comm_.load();
        if (!c) err_ = "the group was aborted (a rank failed)";
        return c;
    }
    // ncclGroupEnd even when a call inside the group failed: an open group would swallow every later call
    bool close_group(bool good) {
        const std::string first = err_;
        const bool closed = ok(api_->GroupEnd(), "ncclGroupEnd");
        if (!good) err_ = first;
        return good && closed;
    }

    RcclApi* api_ = nullptr;
    std::atomic<ncclComm_t> comm_{nullptr};
    std::timed_mutex call_m_;
    ncclUniqueId uid_ = {};
    std::atomic<bool> aborted_{false};
    DevBuf<uint64_t> d_small_;
    PinnedBuf<uint64_t> p_small_;
};

// ---------------------------------------------------------------------------------------------
// ranks as threads of one process
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kMaxLocalWorld = 64;

struct PeerPointers {
    const uint32_t* p[kMaxLocalWorld];
};

__global__ __launch_bounds__(256) void local_reduce_kernel(PeerPointers peers, uint32_t world, size_t n, int op,
                                                           uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t v = peers.p[0][i];
    for (uint32_t k = 1; k < world; ++k) {
        const uint32_t x = peers.p[k][i];
        v = op == 0 ? v + x : op == 1 ? (x < v ? x : v) : (x > v ? x : v);
    }
    out[i] = v;
}

}  // namespace

struct LocalGroup {
    uint32_t world = 1;
    std::mutex m;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> ptr;
    std::vector<std::vector<uint64_t>> counts;
    std::vector<std::vector<uint64_t>> host;
    std::vector<int> device;

    bool aborted = false;               // under m: a rank gave up; nobody waits for anybody any more

    // false = the group was aborted (the rendezvous did not take place)
    bool wait() {
        std::unique_lock<std::mutex> hold(m);
        if (aborted) return false;
        const uint64_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(hold, [&]() { return generation != g || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> hold(m);
        aborted = true;
        cv.notify_all();
    }
};

namespace {

class LocalComm : public Comm {
public:
    LocalComm(LocalGroup* g, uint32_t rank, int device) : g_(g), device_(device) {
        rank_ = rank; world_ = g->world;
        g->device[rank] = device;
    }

    void abort() override { g_->abort(); }

    int host_all_gather(const uint64_t* mine, uint32_t n, uint64_t* all, hipStream_t) override {
        g_->host[rank_].assign(mine, mine + n);
        if (!meet()) return -1;
        for (uint32_t p = 0; p < world_; ++p) memcpy(all + (size_t)p * n, g_->host[p].data(), (size_t)n * 8);
        return meet() ? 0 : -1;
    }

    int all_to_all_v(const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                     size_t elem_bytes, hipStream_t s, bool own_part) override {
        publish(send, s);
        g_->counts[rank_].assign(send_counts, send_counts + world_);
        if (!meet()) return -1;
        size_t ro = 0;
        for (uint32_t p = 0; p < world_; ++p) {
            if (p == rank_ && !own_part) continue;
            size_t so = 0;
            for (uint32_t q = 0; q < rank_; ++q) so += (size_t)g_->counts[p][q] * elem_bytes;
            const size_t bytes = (size_t)g_->counts[p][rank_] * elem_bytes;
            if (g_->counts[p][rank_] != recv_counts[p]) { err_ = "all_to_all_v: counts disagree"; bad_ = true; }
            if (bytes && !bad_ && !copy((char*)recv + ro, (const char*)g_->ptr[p] + so, bytes, p, s)) bad_ = true;
            ro += (size_t)recv_counts[p] * elem_bytes;
        }
        return finish(s);
    }

    int all_gather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
        publish(send, s);
        if (!meet()) return -1;
        for (uint32_t p = 0; p < world_ && !bad_; ++p) {
            if (bytes && !copy((char*)recv + (size_t)p * bytes, g_->ptr[p], bytes, p, s)) bad_ = true;
        }
        return finish(s);
    }

    int all_gather_v(const void* send, void* recv, const uint64_t* counts, size_t elem_bytes, hipStream_t s) override {
        publish(send, s);
        if (!meet()) return -1;
        size_t ro = 0;
        for (uint32_t p = 0; p < world_; ++p) {
            const size_t bytes = (size_t)counts[p] * elem_bytes;
            if (bytes && !bad_ && !copy((char*)recv + ro, g_->ptr[p], bytes, p, s)) bad_ = true;
            ro += bytes;
        }
        return finish(s);
    }

    int all_reduce_u32(uint32_t* buf, size_t n, ReduceOp op, hipStream_t s) override {
        publish(buf, s);
        // (a rank without room for its result still keeps both appointments: the others read its buffer)
        if (tmp_.ensure(n) != hipSuccess) { err_ = "out of memory"; bad_ = true; }
        if (!meet()) return -1;
        if (n && !bad_) {
            PeerPointers pp;
            for (uint32_t p = 0; p < world_; ++p) { pp.p[p] = (const uint32_t*)g_->ptr[p]; peer(p); }
            hipLaunchKernelGGL(local_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pp, world_, n,
                               op == ReduceOp::kSum ? 0 : op == ReduceOp::kMin ? 1 : 2, tmp_.p);
        }
        if (finish(s) != 0) return -1;          // everybody has read everybody's buffer
        if (n && hipMemcpyAsync(buf, tmp_.p, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { err_ = "copy"; return -1; }
        return 0;
    }

    int barrier(hipStream_t s) override {
        bad_ = false;
        if (hipStreamSynchronize(s) != hipSuccess) { err_ = "hipStreamSynchronize"; bad_ = true; }
        if (!meet()) return -1;
        return bad_ ? -1 : 0;
    }

private:
    bool meet() {
        if (g_->wait()) return true;
        err_ = "the group was aborted (a rank failed)";
        return false;
    }
    // a failed rank still takes part in the rendezvous (bad_ is reported at the end of the call)
    void publish(const void* p, hipStream_t s) {
        (void)hipSetDevice(device_);
        bad_ = false;
        // what the peers are about to read must be complete
        if (hipStreamSynchronize(s) != hipSuccess) { err_ = "hipStreamSynchronize"; bad_ = true; }
        g_->ptr[rank_] = p;
    }
    void peer(uint32_t p) {
        const int d = g_->device[p];
        if (d == device_ || (size_t)d >= peer_on_.size() || peer_on_[d]) return;
        (void)hipDeviceEnablePeerAccess(d, 0);  // "already enabled" is fine
        (void)hipGetLastError();
        peer_on_[d] = true;
    }
    bool copy(void* dst, const void* src, size_t bytes, uint32_t from, hipStream_t s) {
        peer(from);
        const hipError_t e = g_->device[from] == device_
                                 ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)
                                 : hipMemcpyPeerAsync(dst, device_, src, g_->device[from], bytes, s);
        if (e != hipSuccess) { err_ = std::string("copy between ranks: ") + hipGetErrorString(e); return false; }
        return true;
    }
    int finish(hipStream_t s) {
        if (hipStreamSynchronize(s) != hipSuccess) { err_ = "hipStreamSynchronize"; bad_ = true; }
        if (!meet()) return -1;                 // the peers' buffers may change from here on
        return bad_ ? -1 : 0;
    }

    LocalGroup* g_;
    int device_;
    bool bad_ = false;
    std::vector<bool> peer_on_ = std::vector<bool>(64, false);
    DevBuf<uint32_t> tmp_;
};

}  // namespace

int rccl_unique_id(void* id, std::string* err) {
    RcclApi* api = rccl_api();
    if (!api->lib) { *err = api->err; return -1; }
    ncclUniqueId uid;
    const ncclResult_t r = api->GetUniqueId(&uid);
    if (r != ncclSuccess) { *err = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return -1; }
    memcpy(id, &uid, sizeof(uid));
    return 0;
}

Comm* create_rccl_comm(uint32_t rank, uint32_t world, const void* id, std::string* err) {
    RcclComm* c = new RcclComm(rank, world);
    if (!c->init(id, err)) { delete c; return nullptr; }
    return c;
}

LocalGroup* create_local_group(uint32_t world) {
    if (world == 0 || world > kMaxLocalWorld) return nullptr;
    LocalGroup* g = new LocalGroup;
    g->world = world;
    g->ptr.assign(world, nullptr);
    g->counts.assign(world, std::vector<uint64_t>());
    g->host.assign(world, std::vector<uint64_t>());
    g->device.assign(world, 0);
    return g;
}

void destroy_local_group(LocalGroup* g) { delete g; }

Comm* create_local_comm(LocalGroup* g, uint32_t rank, int device, std::string* err) {
    if (!g || rank >= g->world) { *err = "bad rank / group"; return nullptr; }
    return new LocalComm(g, rank, device);
}

}  // namespace rala_hip
