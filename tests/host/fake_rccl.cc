// TEST INFRASTRUCTURE, not product: a stand-in for librccl.so that lets the N > 1 paths of csrc/stream_group.hip (chunk offsets, padded tails, root-local
// index, the scatter's send-to-self, the two staging slots under back-to-back pushes) EXECUTE on a box with ONE GPU, where real RCCL refuses two ranks on the
// same device.  It implements exactly the entry points stream_group.hip resolves (names and signatures: /opt/rocm/include/rccl/rccl.h) and nothing else;
// the engine selects it through GSH_RCCL_LIBRARY=<path of this .so> (stream_group.hip: rccl()).  It moves bytes, it measures nothing: any rate taken through
// it is a functional check, never a bandwidth.
//
// Two transports, chosen by how the communicator was made:
//   * ncclCommInitAll (one process, several ranks -- possibly all on the same device): asynchronous and stream-ordered like the real thing.  A transfer is
//     event(sender stream) -> wait on the receiver's stream -> hipMemcpyAsync on the receiver's stream -> event -> wait on the sender's stream, so a caller
//     that re-uses a buffer too early, or reads a result without ordering behind the collective, fails here as it would on xGMI.
//   * ncclCommInitRank with world > 1 (one process per rank): messages are files in a /dev/shm directory named by the unique id (device -> host -> file ->
//     host -> device), synchronous at ncclGroupEnd.  Every rank first posts all it sends, then takes all it receives, so grouped send/recv patterns (the
//     scatter with a send to self) and collectives cannot deadlock.
// Build: hipcc -shared -fPIC -o tests/host/libfake_rccl.so tests/host/fake_rccl.cc   (__graft_entry__.build_fake_rccl)
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace
{
enum
{
    kSuccess = 0,
    kUnhandledHipError = 1,
    kSystemError = 2,
    kInternalError = 3,
    kInvalidArgument = 4,
    kInvalidUsage = 5
};

struct UniqueId
{
    char internal[128];
};

size_t type_bytes(int t)
{
    switch (t)
        {
        case 0:  // ncclInt8 / ncclChar
        case 1:  // ncclUint8
            return 1;
        case 2:  // ncclInt32
        case 3:  // ncclUint32
        case 7:  // ncclFloat32
            return 4;
        case 4:  // ncclInt64
        case 5:  // ncclUint64
        case 8:  // ncclFloat64
            return 8;
        case 6:  // ncclFloat16
        case 9:  // ncclBfloat16
            return 2;
        default:
            return 0;
        }
}

enum Kind
{
    kBroadcast,
    kAllGather,
    kSend,
    kRecv
};

struct Comm;

struct Op
{
    Kind kind;
    Comm* comm;
    const void* send;
    void* recv;
    size_t bytes;  // per-rank count in bytes
    int peer;      // root (broadcast) / peer (send, recv)
    hipStream_t stream;
    uint64_t seq;  // cross-process: message sequence number
    bool done{false};
};

// ranks that live in this process and were made together (ncclCommInitAll, or a communicator of one)
struct Local
{
    std::mutex m;
    int world{0};
    std::vector<Comm*> comms;
    std::deque<Op> pending;  // posted, not yet matched
};

struct Comm
{
    int rank{0};
    int world{1};
    int device{0};
    std::shared_ptr<Local> local;  // in-process transport when set
    std::string dir;               // cross-process transport: mailbox directory
    std::vector<uint64_t> send_seq, recv_seq;
    uint64_t coll_seq{0};
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_group;

struct DeviceGuard
{
    int saved{-1};
    explicit DeviceGuard(int dev)
    {
        (void)hipGetDevice(&saved);
        if (saved != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        int now = -1;
        (void)hipGetDevice(&now);
        if (saved >= 0 && now != saved) (void)hipSetDevice(saved);
    }
};

#define FR_HIP(x)                                     \
    do                                                \
        {                                             \
            if ((x) != hipSuccess)                    \
                {                                     \
                    (void)hipGetLastError();          \
                    return kUnhandledHipError;        \
                }                                     \
        }                                             \
    while (0)

// ------------------------------------------------------------------------------------------------ in-process transport
// src lives on (a, sa), dst on (b, sb): ordered behind everything queued on both streams, and both streams ordered behind the copy
int copy_between(const Comm* a, hipStream_t sa, const void* src, const Comm* b, hipStream_t sb, void* dst, size_t bytes)
{
    if (bytes == 0 || (src == dst && a->device == b->device)) return kSuccess;
    hipEvent_t ready = nullptr, taken = nullptr;
    {
        DeviceGuard g(a->device);
        FR_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        FR_HIP(hipEventRecord(ready, sa));
    }
    {
        DeviceGuard g(b->device);
        FR_HIP(hipStreamWaitEvent(sb, ready, 0));
        FR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, sb));
        FR_HIP(hipEventCreateWithFlags(&taken, hipEventDisableTiming));
        FR_HIP(hipEventRecord(taken, sb));
    }
    {
        DeviceGuard g(a->device);
        FR_HIP(hipStreamWaitEvent(sa, taken, 0));
    }
    (void)hipEventDestroy(ready);  // (released by the runtime once the recorded work has passed)
    (void)hipEventDestroy(taken);
    return kSuccess;
}

// match what can be matched among the posted operations of one in-process group; the caller holds L->m
int progress(Local* L)
{
    bool moved = true;
    while (moved)
        {
            moved = false;
            // point to point: the oldest send a -> b pairs with the oldest recv of b from a
            for (size_t i = 0; i < L->pending.size(); i++)
                {
                    Op& s = L->pending[i];
                    if (s.done || s.kind != kSend) continue;
                    bool earlier = false;  // an older unmatched send of the same pair goes first
                    for (size_t k = 0; k < i; k++)
                        {
                            const Op& o = L->pending[k];
                            if (!o.done && o.kind == kSend && o.comm == s.comm && o.peer == s.peer) earlier = true;
                        }
                    if (earlier) continue;
                    for (size_t j = 0; j < L->pending.size(); j++)
                        {
                            Op& r = L->pending[j];
                            if (r.done || r.kind != kRecv || r.comm->rank != s.peer || r.peer != s.comm->rank) continue;
                            if (r.bytes != s.bytes) return kInvalidArgument;
                            const int rc = copy_between(s.comm, s.stream, s.send, r.comm, r.stream, r.recv, s.bytes);
                            if (rc != kSuccess) return rc;
                            s.done = r.done = true;
                            moved = true;
                            break;
                        }
                }
            // collectives: the oldest unmatched collective of every rank must be the same call
            std::vector<Op*> head(static_cast<size_t>(L->world), nullptr);
            int have = 0;
            for (Op& o : L->pending)
                {
                    if (o.done || (o.kind != kBroadcast && o.kind != kAllGather)) continue;
                    Op*& h = head[static_cast<size_t>(o.comm->rank)];
                    if (h == nullptr)
                        {
                            h = &o;
                            have++;
                        }
                }
            if (have == L->world && L->world > 0)
                {
                    const Op* first = head[0];
                    for (Op* h : head)
                        if (h->kind != first->kind || h->bytes != first->bytes || (h->kind == kBroadcast && h->peer != first->peer)) return kInvalidUsage;
                    if (first->kind == kBroadcast)
                        {
                            const Op* root = head[static_cast<size_t>(first->peer)];
                            for (Op* h : head)
                                {
                                    const int rc = copy_between(root->comm, root->stream, root->send, h->comm, h->stream, h->recv, h->bytes);
                                    if (rc != kSuccess) return rc;
                                }
                        }
                    else
                        for (Op* from : head)
                            for (Op* to : head)
                                {
                                    void* dst = static_cast<char*>(to->recv) + static_cast<size_t>(from->comm->rank) * from->bytes;
                                    const int rc = copy_between(from->comm, from->stream, from->send, to->comm, to->stream, dst, from->bytes);
                                    if (rc != kSuccess) return rc;
                                }
                    for (Op* h : head) h->done = true;
                    moved = true;
                }
            while (!L->pending.empty() && L->pending.front().done) L->pending.pop_front();
        }
    return kSuccess;
}

// ------------------------------------------------------------------------------------------------ cross-process transport
std::string message_path(const Comm* c, char kind, int src, int dst, uint64_t seq)
{
    char name[96];
    std::snprintf(name, sizeof name, "/%c_%d_%d_%llu", kind, src, dst, static_cast<unsigned long long>(seq));
    return c->dir + name;
}

int post_file(const std::string& path, hipStream_t stream, const void* dev_src, size_t bytes)
{
    std::vector<char> host(bytes);
    FR_HIP(hipStreamSynchronize(stream));
    if (bytes) FR_HIP(hipMemcpy(host.data(), dev_src, bytes, hipMemcpyDeviceToHost));
    const std::string tmp = path + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (f == nullptr) return kSystemError;
    const size_t w = bytes ? std::fwrite(host.data(), 1, bytes, f) : 0;
    if (std::fclose(f) != 0 || w != bytes) return kSystemError;
    return std::rename(tmp.c_str(), path.c_str()) == 0 ? kSuccess : kSystemError;
}

double timeout_s()
{
    const char* e = std::getenv("FAKE_RCCL_TIMEOUT_S");
    return e ? std::atof(e) : 120.0;
}

bool wait_for_file(const std::string& path)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path.c_str(), &st) != 0)
        {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    return true;
}

int take_file(const std::string& path, hipStream_t stream, void* dev_dst, size_t bytes, bool remove_it)
{
    if (!wait_for_file(path)) return kSystemError;
    std::vector<char> host(bytes);
    FILE* f = std::fopen(path.c_str(), "rb");
    if (f == nullptr) return kSystemError;
    const size_t r = bytes ? std::fread(host.data(), 1, bytes, f) : 0;
    std::fclose(f);
    if (r != bytes) return kInvalidArgument;  // the two sides disagree about the size
    if (remove_it) (void)std::remove(path.c_str());
    FR_HIP(hipStreamSynchronize(stream));
    if (bytes) FR_HIP(hipMemcpy(dev_dst, host.data(), bytes, hipMemcpyHostToDevice));
    return kSuccess;
}

int run_remote(std::vector<Op>& ops)
{
    // sequence numbers in call order
    for (Op& o : ops)
        {
            Comm* c = o.comm;
            if (o.kind == kSend) o.seq = c->send_seq[static_cast<size_t>(o.peer)]++;
            if (o.kind == kRecv) o.seq = c->recv_seq[static_cast<size_t>(o.peer)]++;
            if (o.kind == kBroadcast || o.kind == kAllGather) o.seq = c->coll_seq++;
        }
    // everything this rank sends ...
    for (Op& o : ops)
        {
            Comm* c = o.comm;
            DeviceGuard g(c->device);
            int rc = kSuccess;
            if (o.kind == kSend) rc = post_file(message_path(c, 'p', c->rank, o.peer, o.seq), o.stream, o.send, o.bytes);
            if (o.kind == kBroadcast && c->rank == o.peer)
                for (int d = 0; d < c->world && rc == kSuccess; d++)
                    if (d != c->rank) rc = post_file(message_path(c, 'b', c->rank, d, o.seq), o.stream, o.send, o.bytes);
            if (o.kind == kAllGather)
                for (int d = 0; d < c->world && rc == kSuccess; d++)
                    if (d != c->rank) rc = post_file(message_path(c, 'g', c->rank, d, o.seq), o.stream, o.send, o.bytes);
            if (rc != kSuccess) return rc;
        }
    // ... then everything it receives
    for (Op& o : ops)
        {
            Comm* c = o.comm;
            DeviceGuard g(c->device);
            int rc = kSuccess;
            if (o.kind == kRecv) rc = take_file(message_path(c, 'p', o.peer, c->rank, o.seq), o.stream, o.recv, o.bytes, true);
            if (o.kind == kBroadcast)
                {
                    if (c->rank != o.peer)
                        rc = take_file(message_path(c, 'b', o.peer, c->rank, o.seq), o.stream, o.recv, o.bytes, true);
                    else if (o.send != o.recv && o.bytes)
                        {
                            FR_HIP(hipMemcpyAsync(o.recv, o.send, o.bytes, hipMemcpyDeviceToDevice, o.stream));
                        }
                }
            if (o.kind == kAllGather)
                for (int s = 0; s < c->world && rc == kSuccess; s++)
                    {
                        void* dst = static_cast<char*>(o.recv) + static_cast<size_t>(s) * o.bytes;
                        if (s != c->rank)
                            rc = take_file(message_path(c, 'g', s, c->rank, o.seq), o.stream, dst, o.bytes, true);
                        else if (dst != o.send && o.bytes)
                            {
                                FR_HIP(hipMemcpyAsync(dst, o.send, o.bytes, hipMemcpyDeviceToDevice, o.stream));
                            }
                    }
            if (rc != kSuccess) return rc;
        }
    return kSuccess;
}

int run(std::vector<Op>& ops)
{
    std::vector<Op> remote;
    for (Op& o : ops)
        {
            if (!o.comm->local)
                {
                    remote.push_back(o);
                    continue;
                }
            Local* L = o.comm->local.get();
            std::lock_guard<std::mutex> lock(L->m);
            L->pending.push_back(o);
        }
    for (Op& o : ops)
        if (o.comm->local)
            {
                Local* L = o.comm->local.get();
                std::lock_guard<std::mutex> lock(L->m);
                const int rc = progress(L);
                if (rc != kSuccess) return rc;
            }
    return remote.empty() ? kSuccess : run_remote(remote);
}

int submit(Op op)
{
    if (op.comm == nullptr) return kInvalidArgument;
    if (g_depth > 0)
        {
            g_group.push_back(op);
            return kSuccess;
        }
    std::vector<Op> one{op};
    return run(one);
}

std::atomic<uint64_t> g_ids{0};
}  // namespace

extern "C"
{
    int fake_rccl_marker = 1;  // (dlsym on this tells a test which library it got)

    int ncclGetVersion(int* v)
    {
        if (v) *v = 99999;  // no real RCCL carries this number
        return kSuccess;
    }

    const char* ncclGetErrorString(int rc)
    {
        switch (rc)
            {
            case kSuccess: return "no error";
            case kUnhandledHipError: return "fake_rccl: unhandled HIP error";
            case kSystemError: return "fake_rccl: system error (mailbox file, or a peer that never posted)";
            case kInternalError: return "fake_rccl: internal error";
            case kInvalidArgument: return "fake_rccl: invalid argument (sizes of the two sides disagree?)";
            case kInvalidUsage: return "fake_rccl: invalid usage (ranks disagree about the collective)";
            default: return "fake_rccl: unknown error";
            }
    }

    int ncclGetUniqueId(UniqueId* id)
    {
        if (id == nullptr) return kInvalidArgument;
        std::memset(id->internal, 0, sizeof id->internal);
        const unsigned long long now = static_cast<unsigned long long>(std::chrono::steady_clock::now().time_since_epoch().count());
        std::snprintf(id->internal, sizeof id->internal, "fake_rccl_%d_%llx_%llu", static_cast<int>(getpid()), now, static_cast<unsigned long long>(g_ids++));
        return kSuccess;
    }

    int ncclCommInitAll(Comm** comms, int ndev, const int* devlist)
    {
        if (comms == nullptr || ndev < 1) return kInvalidArgument;
        auto L = std::make_shared<Local>();
        L->world = ndev;
        for (int i = 0; i < ndev; i++)
            {
                Comm* c = new Comm();
                c->rank = i;
                c->world = ndev;
                c->device = devlist ? devlist[i] : i;
                c->local = L;
                L->comms.push_back(c);
                comms[i] = c;
            }
        return kSuccess;
    }

    int ncclCommInitRank(Comm** comm, int nranks, UniqueId id, int rank)
    {
        if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
        Comm* c = new Comm();
        c->rank = rank;
        c->world = nranks;
        (void)hipGetDevice(&c->device);
        if (nranks == 1)
            {
                c->local = std::make_shared<Local>();
                c->local->world = 1;
                c->local->comms.push_back(c);
                *comm = c;
                return kSuccess;
            }
        id.internal[sizeof id.internal - 1] = '\0';
        for (const char* p = id.internal; *p; p++)
            if (!((*p >= 'a' && *p <= 'z') || (*p >= '0' && *p <= '9') || *p == '_'))
                {
                    delete c;
                    return kInvalidArgument;  // not an id this library made
                }
        c->dir = std::string("/dev/shm/") + id.internal;
        c->send_seq.assign(static_cast<size_t>(nranks), 0);
        c->recv_seq.assign(static_cast<size_t>(nranks), 0);
        (void)mkdir(c->dir.c_str(), 0700);
        char name[64];
        std::snprintf(name, sizeof name, "/join_%d", rank);
        FILE* f = std::fopen((c->dir + name).c_str(), "wb");
        if (f == nullptr)
            {
                delete c;
                return kSystemError;
            }
        std::fclose(f);
        for (int r = 0; r < nranks; r++)  // like the real call: returns once every rank has joined
            {
                std::snprintf(name, sizeof name, "/join_%d", r);
                if (!wait_for_file(c->dir + name))
                    {
                        delete c;
                        return kSystemError;
                    }
            }
        *comm = c;
        return kSuccess;
    }

    int ncclCommDestroy(Comm* c)
    {
        if (c == nullptr) return kSuccess;
        if (!c->dir.empty())
            {
                char name[64];
                std::snprintf(name, sizeof name, "/leave_%d", c->rank);
                FILE* f = std::fopen((c->dir + name).c_str(), "wb");
                if (f) std::fclose(f);
                bool all = true;
                struct stat st;
                for (int r = 0; r < c->world; r++)
                    {
                        std::snprintf(name, sizeof name, "/leave_%d", r);
                        all = all && stat((c->dir + name).c_str(), &st) == 0;
                    }
                if (all)  // the last one out removes the mailbox
                    {
                        for (int r = 0; r < c->world; r++)
                            {
                                std::snprintf(name, sizeof name, "/leave_%d", r);
                                (void)std::remove((c->dir + name).c_str());
                                std::snprintf(name, sizeof name, "/join_%d", r);
                                (void)std::remove((c->dir + name).c_str());
                            }
                        (void)rmdir(c->dir.c_str());
                    }
            }
        delete c;
        return kSuccess;
    }

    int ncclGroupStart()
    {
        g_depth++;
        return kSuccess;
    }

    int ncclGroupEnd()
    {
        if (g_depth <= 0) return kInvalidUsage;
        if (--g_depth > 0) return kSuccess;
        std::vector<Op> ops;
        ops.swap(g_group);
        return run(ops);
    }

    int ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, Comm* comm, hipStream_t stream)
    {
        const size_t sz = type_bytes(datatype);
        if (sz == 0 || comm == nullptr || root < 0 || root >= comm->world) return kInvalidArgument;
        return submit(Op{kBroadcast, comm, sendbuff, recvbuff, count * sz, root, stream, 0});
    }

    int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, Comm* comm, hipStream_t stream)
    {
        const size_t sz = type_bytes(datatype);
        if (sz == 0 || comm == nullptr) return kInvalidArgument;
        return submit(Op{kAllGather, comm, sendbuff, recvbuff, sendcount * sz, 0, stream, 0});
    }

    int ncclSend(const void* sendbuff, size_t count, int datatype, int peer, Comm* comm, hipStream_t stream)
    {
        const size_t sz = type_bytes(datatype);
        if (sz == 0 || comm == nullptr || peer < 0 || peer >= comm->world) return kInvalidArgument;
        return submit(Op{kSend, comm, sendbuff, nullptr, count * sz, peer, stream, 0});
    }

    int ncclRecv(void* recvbuff, size_t count, int datatype, int peer, Comm* comm, hipStream_t stream)
    {
        const size_t sz = type_bytes(datatype);
        if (sz == 0 || comm == nullptr || peer < 0 || peer >= comm->world) return kInvalidArgument;
        return submit(Op{kRecv, comm, nullptr, recvbuff, count * sz, peer, stream, 0});
    }
}
