// gsh_stream_group_*: one IF sample block replicated into N device sample rings over RCCL / xGMI (north_star: "channels
// shard naturally across the 8 GPUs of one node with RCCL broadcast of the shared input sample block").
//
// gnss-sdr feeds every channel from one source through GNU Radio's shared buffers (gnss_flowgraph.cc:1227-1231); with the channels sharded
// over several GPUs that sharing becomes: the block enters ONE GPU (the ingest GPU, rank 0) in the front-end's raw item format (8-bit or
// 16-bit I/Q: 2 or 4 bytes per sample on the wire instead of 8), travels to the others over xGMI, and every GPU converts it to complex64
// into its own ring (gsh_stream).  No reduction exists anywhere in acquisition / tracking, so this replication is the only collective.
//
// xGMI is point to point (one ~153 GB/s link per peer), not a switch:
//   GSH_GROUP_BROADCAST          ncclBroadcast -- simplest; a ring broadcast pushes the whole block through one link per hop;
//   GSH_GROUP_SCATTER_ALLGATHER  the ingest GPU sends a different 1/N of the block to every peer over all its links at once (grouped
//                                ncclSend / ncclRecv), then an all-gather completes it: every link carries block/N bytes twice.
// Two usage models: one process driving several GPUs (gsh_stream_group_create: ncclCommInitAll), or one process per GPU
// (gsh_stream_group_create_rank + an id from gsh_comm_unique_id handed round by whatever launched the ranks).
// RCCL is loaded on first use (dlopen), so that the library itself carries no dependency on it.
#include "sample_convert.h"
#include "sample_stream.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace
{
using gsh::set_error;

// ---- the few RCCL entry points used, resolved at run time (names and signatures: /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct
{
    char internal[128];
} ncclUniqueId;
enum
{
    ncclInt8 = 0
};
struct Rccl
{
    void* lib{nullptr};
    int (*GetUniqueId)(ncclUniqueId*){nullptr};
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int){nullptr};
    int (*CommInitAll)(ncclComm_t*, int, const int*){nullptr};
    int (*CommDestroy)(ncclComm_t){nullptr};
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t){nullptr};
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t){nullptr};
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t){nullptr};
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t){nullptr};
    int (*GroupStart)(){nullptr};
    int (*GroupEnd)(){nullptr};
    const char* (*GetErrorString)(int){nullptr};
    int (*GetVersion)(int*){nullptr};
};

// GSH_RCCL_LIBRARY=<path> names the collective library to load instead of the system's librccl (a site's own RCCL build; the test-only stand-in
// tests/host/libfake_rccl.so that lets the N > 1 paths below run on a one-GPU box).  Read at every group / id creation: a process may switch between
// groups, never while a group made with the other library is alive.
struct RcclSlot
{
    std::mutex m;
    std::string chosen;   // the GSH_RCCL_LIBRARY value the table below was resolved for
    bool resolved{false};
    Rccl table;
    std::string path;     // where the loaded library lives (dladdr)
};

RcclSlot& slot()
{
    static RcclSlot s;
    return s;
}

Rccl* rccl()
{
    RcclSlot& S = slot();
    std::lock_guard<std::mutex> lock(S.m);
    const char* env = std::getenv("GSH_RCCL_LIBRARY");
    const std::string want = env ? env : "";
    Rccl& r = S.table;
    if (!S.resolved || want != S.chosen)
        {
            r = Rccl();
            S.path.clear();
            S.chosen = want;
            S.resolved = true;
            if (!want.empty())
                r.lib = dlopen(want.c_str(), RTLD_NOW | RTLD_LOCAL);  // (named explicitly: no silent fall-back to the system's library)
            else
                for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                    {
                        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                        if (r.lib != nullptr) break;
                    }
            if (r.lib == nullptr) return nullptr;
            auto sym = [&](const char* n) { return dlsym(r.lib, n); };
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
            r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
            r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
            Dl_info info;
            if (r.Broadcast != nullptr && dladdr(reinterpret_cast<void*>(r.Broadcast), &info) != 0 && info.dli_fname != nullptr) S.path = info.dli_fname;
        }
    const bool ok = r.lib && r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.Broadcast && r.AllGather && r.Send && r.Recv && r.GroupStart &&
                    r.GroupEnd;
    return ok ? &r : nullptr;
}

#define GSH_RCCL(call)                                                                                                      \
    do                                                                                                                      \
        {                                                                                                                   \
            const int rc_ = (call);                                                                                         \
            if (rc_ != 0)                                                                                                   \
                return set_error(GSH_ERR_HIP, "RCCL: %s failed: %s", #call, R->GetErrorString ? R->GetErrorString(rc_) : "?"); \
        }                                                                                                                   \
    while (0)
}  // namespace

struct gsh_stream_group
{
    int world{1};                       // ranks in the communicator (= GPUs the block is replicated to)
    int mode{GSH_GROUP_BROADCAST};
    std::vector<int> ranks;             // rank of each LOCAL ring (single process: 0..n-1; one process per GPU: {rank})
    std::vector<gsh_stream*> rings;     // local rings, owned
    std::vector<ncclComm_t> comms;      // one per local ring
    std::vector<void*> stage[2];        // per local ring: two raw-item staging buffers (alternating), padded to world * chunk bytes
    std::vector<void*> piece[2];        // scatter + all-gather: the 1/N this rank receives
    size_t stage_cap{0};                // bytes per staging buffer
    int slot{0};
    bool use_rccl{false};               // world > 1, or a group of one that was asked to go through RCCL all the same (GSH_GROUP_FORCE_RCCL)
    int rccl_ranks{0};                  // ranks of the communicator(s) created (0: none)
    uint64_t rccl_calls{0};             // collectives / point-to-point calls issued
};

namespace
{
size_t padded_bytes(size_t bytes, int world)
{
    const size_t chunk = (bytes + static_cast<size_t>(world) - 1) / static_cast<size_t>(world);
    const size_t chunk16 = (chunk + 15) & ~static_cast<size_t>(15);
    return chunk16 * static_cast<size_t>(world);
}

int ensure_staging(gsh_stream_group* g, size_t bytes)
{
    const size_t need = padded_bytes(bytes, g->world);
    if (need <= g->stage_cap) return GSH_OK;
    for (size_t i = 0; i < g->rings.size(); i++)
        {
            GSH_HIP(hipSetDevice(g->rings[i]->device));
            GSH_HIP(hipStreamSynchronize(g->rings[i]->stream));
            for (int s = 0; s < 2; s++)
                {
                    if (g->stage[s][i]) GSH_HIP(hipFree(g->stage[s][i]));
                    if (g->piece[s][i]) GSH_HIP(hipFree(g->piece[s][i]));
                    g->stage[s][i] = g->piece[s][i] = nullptr;
                    GSH_HIP(hipMalloc(&g->stage[s][i], need));
                    GSH_HIP(hipMalloc(&g->piece[s][i], need / static_cast<size_t>(g->world)));
                }
        }
    g->stage_cap = need;
    return GSH_OK;
}

int group_alloc(gsh_stream_group** out, const int* devices, int n_local, uint64_t capacity, uint32_t max_window)
{
    gsh_stream_group* g = new (std::nothrow) gsh_stream_group();
    GSH_REQUIRE(g != nullptr, "out of host memory");
    for (int i = 0; i < n_local; i++)
        {
            gsh_stream_t* s = nullptr;
            int rc = gsh_stream_create(devices[i], capacity, max_window, &s);
            if (rc != GSH_OK)
                {
                    gsh_stream_group_destroy(g);
                    return rc;
                }
            g->rings.push_back(s);
            g->comms.push_back(nullptr);
            for (int k = 0; k < 2; k++)
                {
                    g->stage[k].push_back(nullptr);
                    g->piece[k].push_back(nullptr);
                }
        }
    *out = g;
    return GSH_OK;
}

// The exchange plan of one rank for one block: the ONLY place where chunk sizes, offsets, peers and the grouping of the calls are decided.  replicate()
// below issues exactly these operations through RCCL; gsh_stream_group_plan hands the same list to whoever wants to check it (tests execute it over
// torch.distributed / gloo on CPU tensors: tests/test_sharding_gloo.py).
int make_plan(uint64_t bytes, int world, int rank, int mode, std::vector<gsh_group_op_t>& ops, uint64_t* padded_out)
{
    ops.clear();
    const uint64_t padded = padded_bytes(static_cast<size_t>(bytes), world);
    const uint64_t chunk = padded / static_cast<uint64_t>(world);
    if (padded_out) *padded_out = padded;
    if (mode == GSH_GROUP_BROADCAST)
        {
            // phase 0: the whole (padded) staging buffer from rank 0, in place
            ops.push_back(gsh_group_op_t{GSH_GROUP_OP_BROADCAST, 0, 0, GSH_GROUP_BUF_STAGE, GSH_GROUP_BUF_STAGE, 0, 0, padded});
            return GSH_OK;
        }
    // phase 0 -- scatter: rank 0 sends piece r to rank r over its link to r (its own piece to itself); everybody receives its piece
    if (rank == 0)
        for (int r = 0; r < world; r++)
            ops.push_back(gsh_group_op_t{GSH_GROUP_OP_SEND, r, 0, GSH_GROUP_BUF_STAGE, GSH_GROUP_BUF_NONE, static_cast<uint64_t>(r) * chunk, 0, chunk});
    ops.push_back(gsh_group_op_t{GSH_GROUP_OP_RECV, 0, 0, GSH_GROUP_BUF_NONE, GSH_GROUP_BUF_PIECE, 0, 0, chunk});
    // phase 1 -- the all-gather of the pieces completes the block everywhere (piece of rank r lands at r * chunk of every staging buffer)
    ops.push_back(gsh_group_op_t{GSH_GROUP_OP_ALLGATHER, 0, 1, GSH_GROUP_BUF_PIECE, GSH_GROUP_BUF_STAGE, 0, 0, chunk});
    return GSH_OK;
}

// queue the replication of `bytes` raw bytes: on the root's local ring the block is in stage[slot][root_local]; afterwards every local
// ring's stage[slot] holds it
int replicate(gsh_stream_group* g, size_t bytes, int slot)
{
    if (!g->use_rccl) return GSH_OK;
    Rccl* R = rccl();
    GSH_REQUIRE(R != nullptr, "RCCL (librccl.so) could not be loaded");
    const size_t n_local = g->rings.size();
    std::vector<std::vector<gsh_group_op_t>> plans(n_local);
    int phases = 0;
    for (size_t i = 0; i < n_local; i++)
        {
            make_plan(bytes, g->world, g->ranks[i], g->mode, plans[i], nullptr);
            for (const gsh_group_op_t& op : plans[i]) phases = std::max(phases, op.phase + 1);
        }
    for (int phase = 0; phase < phases; phase++)
        {
            GSH_RCCL(R->GroupStart());  // (one group per phase: with several local ranks in one thread every call of a phase must be in flight together)
            for (size_t i = 0; i < n_local; i++)
                {
                    GSH_HIP(hipSetDevice(g->rings[i]->device));
                    hipStream_t st = g->rings[i]->stream;
                    auto buf = [&](int which) -> char* {
                        return static_cast<char*>(which == GSH_GROUP_BUF_STAGE ? g->stage[slot][i] : (which == GSH_GROUP_BUF_PIECE ? g->piece[slot][i] : nullptr));
                    };
                    for (const gsh_group_op_t& op : plans[i])
                        {
                            if (op.phase != phase) continue;
                            switch (op.op)
                                {
                                case GSH_GROUP_OP_BROADCAST:
                                    GSH_RCCL(R->Broadcast(buf(op.src_buf) + op.src_offset, buf(op.dst_buf) + op.dst_offset, op.bytes, ncclInt8, op.peer, g->comms[i], st));
                                    break;
                                case GSH_GROUP_OP_SEND:
                                    GSH_RCCL(R->Send(buf(op.src_buf) + op.src_offset, op.bytes, ncclInt8, op.peer, g->comms[i], st));
                                    break;
                                case GSH_GROUP_OP_RECV:
                                    GSH_RCCL(R->Recv(buf(op.dst_buf) + op.dst_offset, op.bytes, ncclInt8, op.peer, g->comms[i], st));
                                    break;
                                case GSH_GROUP_OP_ALLGATHER:
                                    GSH_RCCL(R->AllGather(buf(op.src_buf) + op.src_offset, buf(op.dst_buf) + op.dst_offset, op.bytes, ncclInt8, g->comms[i], st));
                                    break;
                                default:
                                    return set_error(GSH_ERR_STATE, "unknown group operation %d", op.op);
                                }
                            g->rccl_calls++;
                        }
                }
            GSH_RCCL(R->GroupEnd());
        }
    return GSH_OK;
}

bool force_rccl(int mode)
{
    if (mode & GSH_GROUP_FORCE_RCCL) return true;
    const char* e = std::getenv("GSH_GROUP_FORCE_RCCL");
    return e != nullptr && e[0] != '\0' && e[0] != '0';
}

int root_local_index(const gsh_stream_group* g)
{
    for (size_t i = 0; i < g->ranks.size(); i++)
        if (g->ranks[i] == 0) return static_cast<int>(i);
    return -1;
}

int push_common(gsh_stream_group* g, const void* items, bool items_on_device, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
{
    GSH_REQUIRE(g != nullptr && !g->rings.empty(), "null group");
    const size_t isz = gsh::item_bytes(item_type);
    GSH_REQUIRE(isz != 0, "unknown item type %d", item_type);
    for (gsh_stream* s : g->rings)
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
    if (first_index) *first_index = g->rings[0]->next;
    if (n == 0) return GSH_OK;
    const size_t bytes = static_cast<size_t>(n) * isz;
    int rc = ensure_staging(g, bytes);
    if (rc != GSH_OK) return rc;
    const int slot = g->slot;
    g->slot ^= 1;
    const int root = root_local_index(g);
    if (root >= 0)
        {
            GSH_REQUIRE(items != nullptr, "the ingest rank must supply the block");
            gsh_stream* s = g->rings[static_cast<size_t>(root)];
            GSH_HIP(hipSetDevice(s->device));
            GSH_HIP(hipMemcpyAsync(g->stage[slot][static_cast<size_t>(root)], items, bytes, items_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->stream));
        }
    rc = replicate(g, bytes, slot);
    if (rc != GSH_OK) return rc;
    for (size_t i = 0; i < g->rings.size(); i++)
        {
            GSH_HIP(hipSetDevice(g->rings[i]->device));
            rc = gsh::stream_write_device_items(g->rings[i], g->stage[slot][i], n, item_type, inverted_spectrum ? 1 : 0, g->rings[i]->stream);
            if (rc != GSH_OK) return rc;
        }
    return GSH_OK;
}
}  // namespace

extern "C"
{
    int gsh_comm_unique_id(void* id128)
    {
        GSH_REQUIRE(id128 != nullptr, "null id buffer");
        Rccl* R = rccl();
        GSH_REQUIRE(R != nullptr, "RCCL (librccl.so) could not be loaded");
        ncclUniqueId id;
        GSH_RCCL(R->GetUniqueId(&id));
        std::memcpy(id128, id.internal, 128);
        return GSH_OK;
    }

    int gsh_stream_group_create(const int* devices, int n_devices, uint64_t capacity_samples, uint32_t max_window_samples, int mode, gsh_stream_group_t** out)
    {
        GSH_REQUIRE(out != nullptr && devices != nullptr, "null argument");
        *out = nullptr;
        GSH_REQUIRE(n_devices >= 1 && n_devices <= 64, "n_devices %d outside 1..64", n_devices);
        const bool forced = force_rccl(mode);
        mode &= ~GSH_GROUP_FORCE_RCCL;
        GSH_REQUIRE(mode == GSH_GROUP_BROADCAST || mode == GSH_GROUP_SCATTER_ALLGATHER, "unknown mode %d", mode);
        gsh_stream_group* g = nullptr;
        int rc = group_alloc(&g, devices, n_devices, capacity_samples, max_window_samples);
        if (rc != GSH_OK) return rc;
        g->world = n_devices;
        g->mode = mode;
        g->use_rccl = n_devices > 1 || forced;
        for (int i = 0; i < n_devices; i++) g->ranks.push_back(i);
        if (g->use_rccl)
            {
                Rccl* R = rccl();
                if (R == nullptr)
                    {
                        gsh_stream_group_destroy(g);
                        return set_error(GSH_ERR_HIP, "RCCL (librccl.so) could not be loaded");
                    }
                const int e = R->CommInitAll(g->comms.data(), n_devices, devices);
                if (e != 0)
                    {
                        gsh_stream_group_destroy(g);
                        return set_error(GSH_ERR_HIP, "RCCL: ncclCommInitAll failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?");
                    }
                g->rccl_ranks = n_devices;
            }
        *out = g;
        return GSH_OK;
    }

    int gsh_stream_group_create_rank(int device, int rank, int world, const void* id128, uint64_t capacity_samples, uint32_t max_window_samples, int mode,
        gsh_stream_group_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null argument");
        *out = nullptr;
        GSH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d / world %d", rank, world);
        GSH_REQUIRE(world == 1 || id128 != nullptr, "a communicator id is needed for world > 1 (gsh_comm_unique_id on one rank, handed to all)");
        const bool forced = force_rccl(mode);
        mode &= ~GSH_GROUP_FORCE_RCCL;
        GSH_REQUIRE(mode == GSH_GROUP_BROADCAST || mode == GSH_GROUP_SCATTER_ALLGATHER, "unknown mode %d", mode);
        gsh_stream_group* g = nullptr;
        int rc = group_alloc(&g, &device, 1, capacity_samples, max_window_samples);
        if (rc != GSH_OK) return rc;
        g->world = world;
        g->mode = mode;
        g->ranks.push_back(rank);
        g->use_rccl = world > 1 || forced;
        if (g->use_rccl)
            {
                Rccl* R = rccl();
                if (R == nullptr)
                    {
                        gsh_stream_group_destroy(g);
                        return set_error(GSH_ERR_HIP, "RCCL (librccl.so) could not be loaded");
                    }
                ncclUniqueId id;
                if (id128 != nullptr)
                    std::memcpy(id.internal, id128, 128);
                else  // (a forced group of one that was given no id makes its own)
                    {
                        const int e = R->GetUniqueId(&id);
                        if (e != 0)
                            {
                                gsh_stream_group_destroy(g);
                                return set_error(GSH_ERR_HIP, "RCCL: ncclGetUniqueId failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?");
                            }
                    }
                (void)hipSetDevice(device);
                const int e = R->CommInitRank(&g->comms[0], world, id, rank);
                if (e != 0)
                    {
                        gsh_stream_group_destroy(g);
                        return set_error(GSH_ERR_HIP, "RCCL: ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "?");
                    }
                g->rccl_ranks = world;
            }
        *out = g;
        return GSH_OK;
    }

    void gsh_stream_group_destroy(gsh_stream_group_t* g)
    {
        if (g == nullptr) return;
        Rccl* R = rccl();
        for (size_t i = 0; i < g->rings.size(); i++)
            {
                if (g->rings[i] == nullptr) continue;
                (void)hipSetDevice(g->rings[i]->device);
                (void)hipStreamSynchronize(g->rings[i]->stream);
                if (i < g->comms.size() && g->comms[i] != nullptr && R != nullptr) (void)R->CommDestroy(g->comms[i]);
                for (int s = 0; s < 2; s++)
                    {
                        if (i < g->stage[s].size() && g->stage[s][i]) (void)hipFree(g->stage[s][i]);
                        if (i < g->piece[s].size() && g->piece[s][i]) (void)hipFree(g->piece[s][i]);
                    }
                gsh_stream_destroy(g->rings[i]);
            }
        delete g;
    }

    int gsh_stream_group_size(const gsh_stream_group_t* g) { return g ? static_cast<int>(g->rings.size()) : 0; }

    gsh_stream_t* gsh_stream_group_ring(gsh_stream_group_t* g, int local_index)
    {
        if (g == nullptr || local_index < 0 || local_index >= static_cast<int>(g->rings.size())) return nullptr;
        return g->rings[static_cast<size_t>(local_index)];
    }

    int gsh_stream_group_push(gsh_stream_group_t* g, const void* host_items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        return push_common(g, host_items, false, n, item_type, inverted_spectrum, first_index);
    }

    int gsh_stream_group_push_device(gsh_stream_group_t* g, const void* device_items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        return push_common(g, device_items, true, n, item_type, inverted_spectrum, first_index);
    }

    int gsh_stream_group_rccl_info(const gsh_stream_group_t* g, int32_t* rccl_ranks, int32_t* rccl_version, uint64_t* collectives)
    {
        GSH_REQUIRE(g != nullptr, "null group");
        if (rccl_ranks) *rccl_ranks = g->rccl_ranks;
        if (collectives) *collectives = g->rccl_calls;
        if (rccl_version)
            {
                *rccl_version = 0;
                if (g->rccl_ranks > 0)  // (never loads RCCL on behalf of a group that did not use it)
                    {
                        Rccl* R = rccl();
                        int v = 0;
                        if (R != nullptr && R->GetVersion != nullptr && R->GetVersion(&v) == 0) *rccl_version = v;
                    }
            }
        return GSH_OK;
    }

    int gsh_stream_group_plan(uint64_t bytes, int world, int rank, int mode, gsh_group_op_t* ops, int max_ops, int* n_ops, uint64_t* padded_bytes_out)
    {
        GSH_REQUIRE(n_ops != nullptr, "null argument");
        GSH_REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "rank %d / world %d", rank, world);
        mode &= ~GSH_GROUP_FORCE_RCCL;
        GSH_REQUIRE(mode == GSH_GROUP_BROADCAST || mode == GSH_GROUP_SCATTER_ALLGATHER, "unknown mode %d", mode);
        std::vector<gsh_group_op_t> plan;
        make_plan(bytes, world, rank, mode, plan, padded_bytes_out);
        *n_ops = static_cast<int>(plan.size());
        GSH_REQUIRE(ops == nullptr || max_ops >= *n_ops, "the plan has %d operations, room for %d", *n_ops, max_ops);
        if (ops != nullptr) std::copy(plan.begin(), plan.end(), ops);
        return GSH_OK;
    }

    int gsh_comm_library(char* path, int capacity)
    {
        GSH_REQUIRE(path != nullptr && capacity > 0, "null buffer");
        path[0] = '\0';
        Rccl* R = rccl();
        GSH_REQUIRE(R != nullptr, "RCCL (librccl.so, or what GSH_RCCL_LIBRARY names) could not be loaded");
        RcclSlot& S = slot();
        std::lock_guard<std::mutex> lock(S.m);
        std::snprintf(path, static_cast<size_t>(capacity), "%s", S.path.c_str());
        return GSH_OK;
    }

    int gsh_stream_group_wait(gsh_stream_group_t* g)
    {
        GSH_REQUIRE(g != nullptr, "null group");
        for (gsh_stream* s : g->rings)
            {
                GSH_HIP(hipSetDevice(s->device));
                GSH_HIP(hipStreamSynchronize(s->stream));
            }
        return GSH_OK;
    }
}
