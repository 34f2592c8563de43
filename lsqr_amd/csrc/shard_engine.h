// shard_engine.h -- the row-sharded LSQR iteration driven from C++ with RCCL over xGMI
// (included by lsqrhip.hip).
//
// Two ways into the same loop (run_group):
//   * ONE process, ngpu devices: lsqrhip_create_sharded(m, n, nnz, irow, icol, a, ngpu, &h) cuts
//     the rows into ngpu contiguous blocks balanced by nonzeros, builds one sub-handle per device
//     and a communicator over them (ncclCommInitAll); lsqrhip_solve / lsqrhip_aprod /
//     lsqrhip_destroy then work on `h` as on any handle.  This is what the Fortran
//     `initialize(..., ngpu=N)` binds: the drop-in boundary reaches every GPU of the node.
//   * one process PER GPU (bench.py --gpus N under torch.distributed.run): every rank creates
//     its own matrix handle from its row block, lsqrhip_shard_comm_init joins them with an
//     ncclUniqueId that the host passes around, lsqrhip_shard_solve runs the loop.
// The stages are those of shard_api.h; this file adds the four exchanges of an iteration:
//     scalars   all-gather of each rank's 1-3 partial sums + a sum in RANK ORDER by every rank
//               (identical bits everywhere, whatever algorithm RCCL picks for tiny messages)
//     T -> R    the direct reduce-scatter: slice q of rank p's T is sent straight to rank q
//               (ncclSend / ncclRecv inside one group: all 7 xGMI links of a GPU at once; a ring
//               all-reduce of the same 8n bytes is per-link bound), summed there in rank order
//     V         in-place all-gather of the column slices
// RCCL is loaded with dlopen at first use (the library itself does not link it): a process that
// already holds one (PyTorch) shares that copy, and single-GPU users never load it.
#pragma once

#include <dlfcn.h>
#include <memory>
#include <rccl/rccl.h>

namespace lsqrhip {

// sums[j] <- gath[0*msg + j] + gath[1*msg + j] + ... (rank order), j < k; and, msg = SHARD_MSG: the ranks' piece
// maxima of |v| side by side in vmax[P * SHARD_NMAX] (what csb.h's mode-1 product reads instead of a pass over v)
__global__ void k_sum_ranks(const double *__restrict__ gath, int P, int k, int msg, double *__restrict__ sums,
                            double *__restrict__ vmax)
{
    const int j = threadIdx.x;
    if (j < k) {
        double s = gath[j];
        for (int r = 1; r < P; ++r) s = s + gath[msg * r + j];
        sums[j] = s;
    }
    if (vmax != nullptr && msg == SHARD_MSG)
        for (int i = threadIdx.x; i < P * SHARD_NMAX; i += blockDim.x)
            vmax[i] = gath[(i / SHARD_NMAX) * SHARD_MSG + 4 + i % SHARD_NMAX];
}

__global__ void k_max_int(const int *__restrict__ a, int n, int *__restrict__ out)
{
    int m = a[0];
    for (int i = 1; i < n; ++i) m = a[i] > m ? a[i] : m;
    *out = m;
}

}  // namespace lsqrhip

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // (optional: ipc_setup, when a collective of its own failed on this rank)
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;   // (optional: the overlap's second communicator)
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return nullptr;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        if (!p) ok = false;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.CommSplit = (decltype(r.CommSplit))dlsym(r.lib, "ncclCommSplit");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    if (!ok) {
        r.lib = nullptr;
        return nullptr;
    }
    return &r;
}

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess)                                                                     \
            return fail(LSQRHIP_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

static int create_block(int m, int n, int64_t nnz, const int *irow, const int *icol, const double *a, lsqrhip_handle_t *h)
{
    return lsqrhip_create(m, n, nnz, irow, icol, a, h);
}
static int create_block(int m, int n, int64_t nnz, const int *irow, const int *icol, const float *a, lsqrhip_handle_t *h)
{
    return lsqrhip_create_f32(m, n, nnz, irow, icol, a, h);
}

struct ShardRank {
    H *h = nullptr;          // this rank's matrix handle (owned by the group when `owned`)
    int dev = -1;            // its device
    int grank = 0;           // rank in the world
    int64_t row0 = 0;        // first global row of the block
    ncclComm_t comm = nullptr;
    // LSQRHIP_SHARD_OVERLAP=1: the exchanges of the n-vectors run on a stream of their own, part by part, beside the
    // products (a second communicator: one communicator must not be used from two streams at once)
    ncclComm_t comm2 = nullptr;
    hipStream_t cstream = nullptr;
    hipEvent_t evV = nullptr, evRS = nullptr;          // v_q final (compute stream) / all parts of R received (comm stream)
    std::vector<hipEvent_t> evAG, evT;                 // part k of v gathered (comm stream) / part k of T complete (compute stream)
    // exchanges as copies (ShardGroup::loopback): one copy stream per peer (COPY_STREAMS of them, peer mod that), so that
    // the pulls from different peers run on different SDMA engines / links at once; pe[i] = the last copy on ps[i]
    std::vector<hipStream_t> ps;
    std::vector<hipEvent_t> pe;
    hipEvent_t eb = nullptr;                           // "the base stream up to here" for the copy streams to wait on
    // ... and a second set for batches that hang off the EXCHANGE stream (overlapped schedule): with one set the 512-byte
    // pulls of the norms (compute stream) queued on the same copy streams behind that iteration's gather of v -- which
    // the overlap is there to hide (round-4 advisor)
    std::vector<hipStream_t> ps2;
    std::vector<hipEvent_t> pe2;
    hipEvent_t eb2 = nullptr;
    double *T = nullptr, *R = nullptr, *V = nullptr, *sums = nullptr, *gath = nullptr;  // exchange buffers (owned)
    double *xfull = nullptr, *sefull = nullptr, *bloc = nullptr;                         // P*chunk, P*chunk, m_p
};

struct ShardGroup {
    int P = 1;                     // world size
    int m = 0, n = 0;
    int64_t chunk = 0;
    size_t esz = sizeof(double);   // bytes per vector element: 4 for REAL32 handles (T, R, V, x, se, b are float arrays then)
    std::vector<ShardRank> r;      // the ranks driven by this process
    bool owned = false;            // sub-handles belong to the group (single-process form)
    bool loopback = false;         // exchanges by device copies inside this process instead of RCCL (all ranks are local):
                                   // LSQRHIP_SHARD_COPY=1 (peer copies between the devices of a one-process group: no CU
                                   // set aside for an exchange) or the one-device test harness LSQRHIP_SHARD_LOOPBACK=1
    std::vector<hipEvent_t> ev;    // loopback: one event per rank
    bool copy_streams = true;      // ... and one copy stream per peer (LSQRHIP_SHARD_COPY_STREAMS=0 at creation: the pulls on
                                   // the rank's own stream, one after the other)
    int poll_every = 16;
    int overlap = 0;               // LSQRHIP_SHARD_OVERLAP at creation: exchanges in `parts` parts beside the products
    int parts = 1;
    int msg = 4;                   // doubles a rank contributes to the exchange of the norms: 4, or SHARD_MSG when its piece
                                   // maxima of |v_q| ride along (every matrix in column-swept row blocks: csb.h)
    // One process per GPU with LSQRHIP_SHARD_COPY=1 ("ipc", round 5): the n-vector exchanges are PULLS by the copy engines
    // from the peers' buffers, which hipIpcOpenMemHandle maps into this process -- no RCCL send / receive kernel takes CUs
    // from sweeps that hold every CU and all of its LDS.  RCCL keeps the all-gather of the norms (<= 512 bytes per rank:
    // a one-workgroup kernel that is also the fence in front of the pulls of v) and an 8-byte all-gather as the fence in
    // front of the pulls of T.  peer*[p]: rank p's T, V, xfull, sefull as seen from here (own entry: the local buffer).
    bool ipc = false;
    std::vector<char *> peerT, peerV, peerX, peerSE;
    std::vector<void *> ipc_opened;   // what hipIpcCloseMemHandle must see again
    double *bar = nullptr;            // [1 + P] the fence's all-gather
    double *bar2 = nullptr;           // ... and the one of the exchange stream (overlapped schedule)
    // one captured batch of `poll_every` iterations -- stages AND exchanges of every local rank (capture_group_batch)
    hipGraphExec_t gexec = nullptr;
    std::vector<int> gexec_epoch;  // the ranks' graph_epoch at capture
    int graph_state = 0;           // 0 not tried, 1 in use, -1 capture failed once: eager launches from then on
};

static H *lsqrhip_group_rank0(H *h) { return h->group->r[0].h; }
// option "shard_overlap" / "shard_parts": what the group this handle belongs to runs (0 / 1: no group)
static int64_t shard_effective(const H *h, int what)   // 0 overlap | 1 parts | 2 exchanges as copies
{
    const ShardGroup *g = h->group ? h->group : h->mp;
    if (g == nullptr) return what == 1 ? 1 : 0;
    if (what == 2) return (g->ipc || (g->loopback && g->P > 1)) ? 1 : 0;
    return what == 1 ? (g->overlap ? g->parts : 1) : g->overlap;
}
static H *log_owner(H *h) { return h->group && !h->group->r.empty() && h->group->r[0].h ? h->group->r[0].h : h; }

static void free_group(ShardGroup *g)
{
    if (!g) return;
    if (g->gexec) (void)hipGraphExecDestroy(g->gexec);
    if (!g->r.empty() && (g->bar || !g->ipc_opened.empty())) {
        (void)hipSetDevice(g->r[0].dev >= 0 ? g->r[0].dev : 0);
        (void)hipDeviceSynchronize();   // (no copy of ours may still read a peer's buffer)
        for (void *p : g->ipc_opened) (void)hipIpcCloseMemHandle(p);
        if (g->bar) (void)hipFree(g->bar);
        if (g->bar2) (void)hipFree(g->bar2);
    }

    for (size_t i = 0; i < g->ev.size(); ++i) {
        (void)hipSetDevice(g->r[i].dev >= 0 ? g->r[i].dev : 0);
        if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
    }
    for (ShardRank &k : g->r) {
        if (k.h) (void)hipSetDevice(k.h->device);
        else if (k.dev >= 0) (void)hipSetDevice(k.dev);
        if (k.comm2 && rccl()) (void)rccl()->CommDestroy(k.comm2);
        if (k.comm && rccl()) (void)rccl()->CommDestroy(k.comm);
        for (hipEvent_t e : k.evAG) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : k.evT) if (e) (void)hipEventDestroy(e);
        if (k.evV) (void)hipEventDestroy(k.evV);
        if (k.evRS) (void)hipEventDestroy(k.evRS);
        if (k.cstream) (void)hipStreamDestroy(k.cstream);
        for (hipEvent_t e : k.pe) if (e) (void)hipEventDestroy(e);
        for (hipStream_t st : k.ps) if (st) (void)hipStreamDestroy(st);
        if (k.eb) (void)hipEventDestroy(k.eb);
        for (hipEvent_t e : k.pe2) if (e) (void)hipEventDestroy(e);
        for (hipStream_t st : k.ps2) if (st) (void)hipStreamDestroy(st);
        if (k.eb2) (void)hipEventDestroy(k.eb2);
        for (double *p : {k.T, k.R, k.V, k.sums, k.gath, k.xfull, k.sefull, k.bloc})
            if (p) (void)hipFree(p);
        if (g->owned && k.h) lsqrhip_destroy(k.h);
    }
    delete g;
}

static void release_groups(H *h)
{
    if (h->mp) {
        h->mp->r[0].h = nullptr;  // this very handle: not the group's to destroy
        free_group(h->mp);
        h->mp = nullptr;
    }
    if (h->group) {
        free_group(h->group);
        h->group = nullptr;
    }
}

static int alloc_rank_buffers(ShardGroup &g, ShardRank &k)
{
    k.dev = k.h->device;
    HIPCHK(hipSetDevice(k.h->device));
    g.esz = k.h->f32 ? sizeof(float) : sizeof(double);
    const size_t full = (size_t)std::max<int64_t>(g.chunk * g.P, 1);
    for (double **pp : {&k.T, &k.R, &k.V, &k.xfull, &k.sefull}) HIPCHK(hipMalloc((void **)pp, g.esz * full));
    HIPCHK(hipMalloc((void **)&k.sums, sizeof(double) * SHARD_MSG));
    HIPCHK(hipMalloc((void **)&k.gath, sizeof(double) * SHARD_MSG * (size_t)g.P));
    HIPCHK(hipMalloc((void **)&k.bloc, g.esz * (size_t)std::max(k.h->m, 1)));
    return LSQRHIP_OK;
}

// ---- the exchanges -------------------------------------------------------------------------------
// Loopback (LSQRHIP_SHARD_LOOPBACK=1, single-process groups only; a TEST HARNESS, never chosen by itself): the same
// three exchanges as device-to-device copies between the ranks' buffers, for nodes with fewer GPUs than ranks --
// several ranks then share a device, which RCCL refuses.  Everything but the RCCL calls themselves is the
// production path: stages, buffers, slices, order of the sums.
static int fence_ranks(ShardGroup &g)  // every rank's stream waits for all that is queued on every other rank's
{
    if (g.ev.empty()) {
        g.ev.assign(g.r.size(), nullptr);
        for (size_t i = 0; i < g.r.size(); ++i) {
            HIPCHK(hipSetDevice(g.r[i].h->device));
            HIPCHK(hipEventCreateWithFlags(&g.ev[i], hipEventDisableTiming));
        }
    }
    for (size_t i = 0; i < g.r.size(); ++i) {
        HIPCHK(hipSetDevice(g.r[i].h->device));
        HIPCHK(hipEventRecord(g.ev[i], g.r[i].h->stream));
    }
    for (size_t i = 0; i < g.r.size(); ++i) {
        HIPCHK(hipSetDevice(g.r[i].h->device));
        for (size_t j = 0; j < g.r.size(); ++j)
            if (j != i) HIPCHK(hipStreamWaitEvent(g.r[i].h->stream, g.ev[j], 0));
    }
    return LSQRHIP_OK;
}

// Exchanges as copies: rank `dst` PULLS from its peers.  A batch of pulls hangs off a base stream of dst (its compute
// stream, or the exchange stream of the overlapped schedule): the copy streams wait for the base as it stands at
// pull_begin (whatever the base had waited for -- the sources' data -- they inherit) and, per pull, for an event of the
// source if given; pull_end makes the base wait for every copy.  One copy stream per peer (mod COPY_STREAMS): pulls from
// different peers are independent commands -- on a node they go over different links at once, which one stream (one
// copy at a time: one link at a time) would not allow.  LSQRHIP_SHARD_COPY_STREAMS=0: all pulls on the base itself.
constexpr int COPY_STREAMS = 8;
struct PullBatch {
    ShardRank *q = nullptr;
    hipStream_t base = nullptr;
    bool streams = false;
    unsigned used = 0;   // copy streams this batch touched
    std::vector<hipStream_t> *ps = nullptr;   // the set this batch uses (the rank's, or its exchange stream's)
    std::vector<hipEvent_t> *pe = nullptr;
    hipEvent_t *eb = nullptr;
};
static int pull_begin(PullBatch &b, const ShardGroup &g, ShardRank &dst, hipStream_t base)
{
    b.q = &dst;
    b.base = base;
    b.streams = g.copy_streams;
    b.used = 0;
    HIPCHK(hipSetDevice(dst.h->device));
    if (!b.streams) return LSQRHIP_OK;
    const bool second = dst.cstream != nullptr && base == dst.cstream;
    b.ps = second ? &dst.ps2 : &dst.ps;
    b.pe = second ? &dst.pe2 : &dst.pe;
    b.eb = second ? &dst.eb2 : &dst.eb;
    if (b.ps->empty()) {
        b.ps->assign(COPY_STREAMS, nullptr);
        b.pe->assign(COPY_STREAMS, nullptr);
        for (int i = 0; i < COPY_STREAMS; ++i) {
            HIPCHK(hipStreamCreateWithFlags(&(*b.ps)[(size_t)i], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&(*b.pe)[(size_t)i], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(b.eb, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(*b.eb, base));
    return LSQRHIP_OK;
}
static int pull(PullBatch &b, const ShardRank &src, void *d, const void *s, size_t bytes, hipEvent_t src_ready = nullptr)
{
    if (!bytes) return LSQRHIP_OK;
    ShardRank &q = *b.q;
    hipStream_t st = b.base;
    if (b.streams) {
        const int i = src.grank % COPY_STREAMS;
        st = (*b.ps)[(size_t)i];
        if (!(b.used & (1u << i))) HIPCHK(hipStreamWaitEvent(st, *b.eb, 0));
        b.used |= 1u << i;
    }
    if (src_ready) HIPCHK(hipStreamWaitEvent(st, src_ready, 0));
    if (src.h->device != q.h->device)
        HIPCHK(hipMemcpyPeerAsync(d, q.h->device, s, src.h->device, bytes, st));
    else
        HIPCHK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, st));
    return LSQRHIP_OK;
}
static int pull_end(PullBatch &b)
{
    if (!b.streams) return LSQRHIP_OK;
    ShardRank &q = *b.q;
    (void)q;
    for (int i = 0; i < COPY_STREAMS; ++i)
        if (b.used & (1u << i)) {
            HIPCHK(hipEventRecord((*b.pe)[(size_t)i], (*b.ps)[(size_t)i]));
            HIPCHK(hipStreamWaitEvent(b.base, (*b.pe)[(size_t)i], 0));
        }
    return LSQRHIP_OK;
}
// (one process per GPU with copies over IPC-mapped buffers: defined behind the exchanges)
static int ipc_fence(ShardGroup &g);
static int ipc_fence2(ShardGroup &g);   // ... on the exchange stream and its communicator (overlapped schedule)
static int ipc_pull(PullBatch &b, int p, void *d, const void *s, size_t bytes);
// element i of a vector buffer (binary64 or REAL32 elements)
static inline char *at(double *base, size_t i, size_t esz) { return reinterpret_cast<char *>(base) + i * esz; }

// sums[0..k) <- sum over ranks, in rank order, same bits everywhere.  `with_v`: the in-place all-gather of the v
// slices rides in the same RCCL group (one collective latency instead of two: after ST_VCOMBINE the slices are
// final -- the update that follows only reads them).
// `fused`: the scalar step that follows sums the ranks itself (shard_api.h k_shard_s1g / k_shard_s2g): no k_sum_ranks
static int ex_scalars(ShardGroup &g, int k, bool with_v = false, bool fused = false)
{
    Rccl *rc = rccl();
    const size_t c = (size_t)g.chunk;
    const size_t msg = (size_t)g.msg;
    const ncclDataType_t vtype = g.esz == sizeof(float) ? ncclFloat : ncclDouble;
    auto sum_ranks = [&](ShardRank &q, const double *gath) -> int {
        HIPCHK(hipSetDevice(q.h->device));
        hipLaunchKernelGGL(k_sum_ranks, dim3(1), dim3(64), 0, q.h->stream, gath, g.P, k, g.msg, q.sums,
                           g.msg == SHARD_MSG ? q.h->xmax_part : (double *)nullptr);
        return LSQRHIP_OK;
    };
    if (g.P > 1 && g.loopback) {
        RET(fence_ranks(g));
        for (ShardRank &q : g.r) {
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.h->stream));
            for (ShardRank &p : g.r) {
                RET(pull(pb, p, q.gath + msg * (size_t)p.grank, p.sums, msg * sizeof(double)));
                if (with_v && &p != &q)
                    RET(pull(pb, p, at(q.V, (size_t)p.grank * c, g.esz), at(p.V, (size_t)p.grank * c, g.esz), c * g.esz));
            }
            RET(pull_end(pb));
        }
        RET(fence_ranks(g));
        if (!fused)
            for (ShardRank &q : g.r) RET(sum_ranks(q, q.gath));
        return LSQRHIP_OK;
    }
    if (g.P > 1 && g.ipc) {
        // the norms by RCCL (one tiny kernel) -- which, complete on this rank, says every rank's v_q is final: the
        // slices are then PULLED by the copy engines, every peer's at once.  (What follows reads v only after the pulls:
        // pull_end makes the stream wait.  The peers' next write of v_q lies behind the next exchange of norms.)
        ShardRank &q = g.r[0];
        NCCLCHK(rc->AllGather(q.sums, q.gath, msg, ncclDouble, q.comm, q.h->stream));
        if (with_v && c > 0) {
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.h->stream));
            for (int p = 0; p < g.P; ++p)
                if (p != q.grank)
                    RET(ipc_pull(pb, p, at(q.V, (size_t)p * c, g.esz), g.peerV[(size_t)p] + (size_t)p * c * g.esz, c * g.esz));
            RET(pull_end(pb));
        }
        if (!fused) RET(sum_ranks(q, q.gath));
        return LSQRHIP_OK;
    }
    if (g.P > 1) {
        NCCLCHK(rc->GroupStart());
        for (ShardRank &q : g.r) {
            NCCLCHK(rc->AllGather(q.sums, q.gath, msg, ncclDouble, q.comm, q.h->stream));
            if (with_v && c > 0)
                NCCLCHK(rc->AllGather(at(q.V, (size_t)q.grank * c, g.esz), q.V, c, vtype, q.comm, q.h->stream));
        }
        NCCLCHK(rc->GroupEnd());
        if (!fused)
            for (ShardRank &q : g.r) RET(sum_ranks(q, q.gath));
        return LSQRHIP_OK;
    }
    // a world of one: nothing to exchange, but the piece maxima still go where mode 1 looks for them
    if (g.msg == SHARD_MSG && k == 2 && !fused)
        for (ShardRank &q : g.r) RET(sum_ranks(q, q.sums));
    return LSQRHIP_OK;
}

static int ex_scatter(ShardGroup &g)  // slice q of every rank's T -> rank q's R[rank]
{
    Rccl *rc = rccl();
    const size_t c = (size_t)g.chunk, e = g.esz;
    const ncclDataType_t vtype = e == sizeof(float) ? ncclFloat : ncclDouble;
    if (g.P == 1 || c == 0) return LSQRHIP_OK;   // (the rank's own slice is read in T where it lies: shard_api.h own_in_T)
    if (g.loopback) {
        RET(fence_ranks(g));
        for (ShardRank &q : g.r) {
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.h->stream));
            for (ShardRank &p : g.r)
                if (&p != &q) RET(pull(pb, p, at(q.R, (size_t)p.grank * c, e), at(p.T, (size_t)q.grank * c, e), c * e));
            RET(pull_end(pb));
        }
        return fence_ranks(g);
    }
    if (g.ipc) {   // every rank's T complete (the fence), then slice q of every peer's T pulled into R[peer]
        ShardRank &q = g.r[0];
        RET(ipc_fence(g));
        PullBatch pb;
        RET(pull_begin(pb, g, q, q.h->stream));
        for (int p = 0; p < g.P; ++p)
            if (p != q.grank)
                RET(ipc_pull(pb, p, at(q.R, (size_t)p * c, e), g.peerT[(size_t)p] + (size_t)q.grank * c * e, c * e));
        return pull_end(pb);   // (the peers overwrite T behind the next exchange of norms: every pull is done by then)
    }
    // the rank's own slice never leaves the device (nor T: k_rs_combine reads it there); the others go to their
    // owners over all links at once
    NCCLCHK(rc->GroupStart());
    for (ShardRank &q : g.r)
        for (int peer = 0; peer < g.P; ++peer) {
            if (peer == q.grank) continue;
            NCCLCHK(rc->Send(at(q.T, (size_t)peer * c, e), c, vtype, peer, q.comm, q.h->stream));
            NCCLCHK(rc->Recv(at(q.R, (size_t)peer * c, e), c, vtype, peer, q.comm, q.h->stream));
        }
    NCCLCHK(rc->GroupEnd());
    return LSQRHIP_OK;
}

static int ex_gather(ShardGroup &g, bool x_too, bool se_too)  // in-place all-gather of the column slices
{
    Rccl *rc = rccl();
    const size_t c = (size_t)g.chunk, e = g.esz;
    const ncclDataType_t vtype = e == sizeof(float) ? ncclFloat : ncclDouble;
    if (g.P == 1 || c == 0) return LSQRHIP_OK;
    if (g.loopback) {
        RET(fence_ranks(g));
        for (ShardRank &q : g.r) {
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.h->stream));
            for (ShardRank &p : g.r) {
                if (&p == &q) continue;
                const size_t o = (size_t)p.grank * c;
                if (!x_too) RET(pull(pb, p, at(q.V, o, e), at(p.V, o, e), c * e));
                if (x_too) RET(pull(pb, p, at(q.xfull, o, e), at(p.xfull, o, e), c * e));
                if (se_too) RET(pull(pb, p, at(q.sefull, o, e), at(p.sefull, o, e), c * e));
            }
            RET(pull_end(pb));
        }
        return fence_ranks(g);
    }
    if (g.ipc) {   // fence, pulls, fence: what follows (the next solve's first stage) may overwrite the sources
        ShardRank &q = g.r[0];
        RET(ipc_fence(g));
        PullBatch pb;
        RET(pull_begin(pb, g, q, q.h->stream));
        for (int p = 0; p < g.P; ++p) {
            if (p == q.grank) continue;
            const size_t o = (size_t)p * c;
            if (!x_too) RET(ipc_pull(pb, p, at(q.V, o, e), g.peerV[(size_t)p] + o * e, c * e));
            if (x_too) RET(ipc_pull(pb, p, at(q.xfull, o, e), g.peerX[(size_t)p] + o * e, c * e));
            if (se_too) RET(ipc_pull(pb, p, at(q.sefull, o, e), g.peerSE[(size_t)p] + o * e, c * e));
        }
        RET(pull_end(pb));
        return ipc_fence(g);
    }
    NCCLCHK(rc->GroupStart());
    for (ShardRank &q : g.r) {
        const size_t o = (size_t)q.grank * c;
        if (!x_too) NCCLCHK(rc->AllGather(at(q.V, o, e), q.V, c, vtype, q.comm, q.h->stream));
        if (x_too) NCCLCHK(rc->AllGather(at(q.xfull, o, e), q.xfull, c, vtype, q.comm, q.h->stream));
        if (se_too) NCCLCHK(rc->AllGather(at(q.sefull, o, e), q.sefull, c, vtype, q.comm, q.h->stream));
    }
    NCCLCHK(rc->GroupEnd());
    return LSQRHIP_OK;
}

// ---- one process per GPU, exchanges as copies over IPC-mapped buffers (ShardGroup::ipc) ---------------------------
// `when every rank's stream has come this far, go on`: an 8-byte all-gather (one workgroup for a few microseconds)
static int ipc_fence(ShardGroup &g)
{
    ShardRank &q = g.r[0];
    NCCLCHK(rccl()->AllGather(g.bar, g.bar + 1, 1, ncclDouble, q.comm, q.h->stream));
    return LSQRHIP_OK;
}
static int ipc_fence2(ShardGroup &g)
{
    ShardRank &q = g.r[0];
    NCCLCHK(rccl()->AllGather(g.bar2, g.bar2 + 1, 1, ncclDouble, q.comm2, q.cstream));
    return LSQRHIP_OK;
}
// this rank pulls `bytes` from peer `p`'s buffer on the copy stream of that peer (PullBatch: all peers at once)
static int ipc_pull(PullBatch &b, int p, void *d, const void *s, size_t bytes)
{
    if (!bytes) return LSQRHIP_OK;
    ShardRank &q = *b.q;
    hipStream_t st = b.base;
    (void)q;
    if (b.streams) {
        const int i = p % COPY_STREAMS;
        st = (*b.ps)[(size_t)i];
        if (!(b.used & (1u << i))) HIPCHK(hipStreamWaitEvent(st, *b.eb, 0));
        b.used |= 1u << i;
    }
    HIPCHK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, st));
    return LSQRHIP_OK;
}
// Maps every peer's T, V, xfull, sefull into this process.  Collective: every rank calls it; the handles travel by an
// all-gather on the communicator, and so does the verdict -- the ranks switch to copies together or not at all.
static int ipc_setup(ShardGroup &g)
{
    ShardRank &q = g.r[0];
    Rccl *rc = rccl();
    const int P = g.P, me = q.grank;
    constexpr int NB = 4;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64-byte records");
    // A COLLECTIVE: between its two all-gathers no rank may leave.  Every local failure (an allocation, a handle that
    // cannot be made or opened) is folded into `ok`, which travels with the records: a rank that failed still runs both
    // exchanges, with ok = 0, and all ranks decline together.  Only a failure of an exchange ITSELF (the all-gather or
    // the copies around it) ends the set-up: the peers may be waiting inside a collective this rank never joined, so
    // the communicator is aborted before the error is returned (ncclCommAbort, where the library has it).
    int ok = 1;
    std::string why;
    auto soft = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) {
            if (ok) why = std::string(what) + ": " + hipGetErrorString(e);
            ok = 0;
            (void)hipGetLastError();
        }
    };
    soft(hipSetDevice(q.h->device), "hipSetDevice");
    soft(hipMalloc((void **)&g.bar, sizeof(double) * (size_t)(1 + P)), "hipMalloc(bar)");
    if (g.bar) soft(hipMemset(g.bar, 0, sizeof(double) * (size_t)(1 + P)), "hipMemset(bar)");
    soft(hipMalloc((void **)&g.bar2, sizeof(double) * (size_t)(1 + P)), "hipMalloc(bar2)");
    if (g.bar2) soft(hipMemset(g.bar2, 0, sizeof(double) * (size_t)(1 + P)), "hipMemset(bar2)");
    std::vector<hipIpcMemHandle_t> mine(NB);
    double *bufs[NB] = {q.T, q.V, q.xfull, q.sefull};
    for (int i = 0; i < NB; ++i) soft(hipIpcGetMemHandle(&mine[(size_t)i], bufs[i]), "hipIpcGetMemHandle");
    const size_t recsz = 64 * NB + 64;
    char *d_send = nullptr, *d_recv = nullptr;
    hipError_t e_alloc = hipMalloc((void **)&d_send, recsz);
    if (e_alloc == hipSuccess) e_alloc = hipMalloc((void **)&d_recv, recsz * (size_t)P);
    std::vector<char> rec(recsz, 0), recs(recsz * (size_t)P);
    std::memcpy(rec.data(), mine.data(), 64 * NB);
    // one exchange: every step is attempted in order, the first failure is kept (no early return: the scratch buffers
    // are released in one place below)
    auto exchange = [&]() -> int {
        if (e_alloc != hipSuccess) return fail(LSQRHIP_ERR_ALLOC, std::string("ipc_setup scratch: ") + hipGetErrorString(e_alloc));
        rec[64 * NB] = (char)ok;
        hipError_t e = hipMemcpy(d_send, rec.data(), rec.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(LSQRHIP_ERR_HIP, std::string("ipc_setup upload: ") + hipGetErrorString(e));
        ncclResult_t nr = rc->AllGather(d_send, d_recv, rec.size(), ncclChar, q.comm, q.h->stream);
        if (nr != ncclSuccess)
            return fail(LSQRHIP_ERR_HIP, std::string("ipc_setup ncclAllGather: ") + (rc->GetErrorString ? rc->GetErrorString(nr) : "?"));
        e = hipStreamSynchronize(q.h->stream);
        if (e == hipSuccess) e = hipMemcpy(recs.data(), d_recv, recs.size(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(LSQRHIP_ERR_HIP, std::string("ipc_setup download: ") + hipGetErrorString(e));
        for (int p = 0; p < P; ++p) ok = ok && recs[(size_t)p * rec.size() + 64 * NB] != 0;
        return LSQRHIP_OK;
    };
    int rcx = exchange();
    if (rcx == LSQRHIP_OK) {
        g.peerT.assign((size_t)P, nullptr);
        g.peerV.assign((size_t)P, nullptr);
        g.peerX.assign((size_t)P, nullptr);
        g.peerSE.assign((size_t)P, nullptr);
        std::vector<char *> *dst[NB] = {&g.peerT, &g.peerV, &g.peerX, &g.peerSE};
        for (int p = 0; p < P && ok; ++p)
            for (int i = 0; i < NB; ++i) {
                if (p == me) {
                    (*dst[i])[(size_t)p] = reinterpret_cast<char *>(bufs[i]);
                    continue;
                }
                hipIpcMemHandle_t hd;
                std::memcpy(&hd, recs.data() + (size_t)p * rec.size() + 64 * i, 64);
                void *ptr = nullptr;
                if (hipIpcOpenMemHandle(&ptr, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess || ptr == nullptr) {
                    ok = 0;
                    (void)hipGetLastError();
                    break;
                }
                g.ipc_opened.push_back(ptr);
                (*dst[i])[(size_t)p] = reinterpret_cast<char *>(ptr);
            }
        // the verdict: every rank must have every buffer of every peer
        rcx = exchange();
    }
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (rcx != LSQRHIP_OK) {   // an exchange failed on THIS rank: do not leave the peers waiting in it
        const std::string keep = g_last_error;
        if (rc->CommAbort && q.comm) {
            (void)rc->CommAbort(q.comm);
            q.comm = nullptr;
        }
        g_last_error = keep;
        return rcx;
    }
    g.ipc = ok != 0;
    g.copy_streams = env_int("LSQRHIP_SHARD_COPY_STREAMS", 1) != 0;
    return LSQRHIP_OK;
}

static int stage_all(ShardGroup &g, int st, int phase = -1)
{
    for (ShardRank &q : g.r) RET(shard_stage_phase(q.h, st, phase));
    return LSQRHIP_OK;
}

// ---- LSQRHIP_SHARD_OVERLAP=1: the two n-vector exchanges of an iteration in parts, beside the products -------------
// Every slice travels in G parts (lsqrhip.hip shard_part).  The layouts of a rank's blocks were built for it
// (finish_create): mode 1 sweeps part k of all slices in its phase k, so it only needs part k of the all-gather of v
// -- part k + 1 arrives while it runs; mode 2 completes part k of all slices of T in its phase k, which is sent
// while phase k + 1 runs.  Compute stream S = the handle's; exchange stream C = cstream.  Per iteration:
//     S: wait evAG[k]; mode 1 phase k          (k = 0 .. G-1)        C: (still gathering parts k+1 .. of the last v)
//     S: beta (all-gather of the norms, on S, first communicator)
//     S: mode 2 phase k; record evT[k]         (k = 0 .. G-1)        C: wait evT[k]; part k of T -> the owners' R
//     S: wait evRS; v_q <- combine(R); record evV                    C: record evRS;  wait evV; gather part k of v,
//     S: alpha (norms + piece maxima); x_q, w_q update                  record evAG[k]   (k = 0 .. G-1)
// Nothing of the arithmetic changes: the same kernels on the same data in the same order of sums -- a solve with the
// switch on is bit for bit the solve with it off on the same layouts (tests/test_gpu_engine.py).
static int overlap_setup(ShardGroup &g)
{
    for (ShardRank &q : g.r) {
        if (q.cstream) continue;
        HIPCHK(hipSetDevice(q.h->device));
        {   // (highest priority: the sweeps hold every CU with one LDS-filling workgroup each, so a send / receive
            // kernel only finds room where a sweep's workgroup has just finished -- it should be first in line there)
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIPCHK(hipStreamCreateWithPriority(&q.cstream, hipStreamNonBlocking, hi));
        }
        HIPCHK(hipEventCreateWithFlags(&q.evV, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&q.evRS, hipEventDisableTiming));
        q.evAG.assign((size_t)g.parts, nullptr);
        q.evT.assign((size_t)g.parts, nullptr);
        for (int k = 0; k < g.parts; ++k) {
            HIPCHK(hipEventCreateWithFlags(&q.evAG[(size_t)k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&q.evT[(size_t)k], hipEventDisableTiming));
        }
    }
    return LSQRHIP_OK;
}

// part k of slice `sl`: offset from the slice's start and length, in elements
static void part_of(const ShardGroup &g, int sl, int k, size_t *off, size_t *len)
{
    int64_t lo, hi;
    shard_part(g.n, g.P, g.parts, sl, k, &lo, &hi);
    *off = (size_t)(lo - std::min<int64_t>((int64_t)sl * g.chunk, g.n));
    *len = (size_t)(hi - lo);
}

// part k of T to the owners, on the exchange streams; waits for evT[k]
static int ov_scatter_part(ShardGroup &g, int k)
{
    Rccl *rc = rccl();
    const size_t c = (size_t)g.chunk, e = g.esz;
    const ncclDataType_t vtype = e == sizeof(float) ? ncclFloat : ncclDouble;
    if (g.loopback) {
        for (ShardRank &q : g.r) {
            size_t off, len;
            part_of(g, q.grank, k, &off, &len);
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.cstream));
            for (ShardRank &p : g.r) {
                if (&p == &q || len == 0) continue;
                RET(pull(pb, p, at(q.R, (size_t)p.grank * c + off, e), at(p.T, (size_t)q.grank * c + off, e), len * e,
                         p.evT[(size_t)k]));
            }
            RET(pull_end(pb));
        }
        return LSQRHIP_OK;
    }
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        HIPCHK(hipStreamWaitEvent(q.cstream, q.evT[(size_t)k], 0));
    }
    if (g.ipc) {   // every rank's phase k of mode 2 done (the fence on the exchange stream), then part k of my slice pulled
        ShardRank &q = g.r[0];
        size_t off, len;
        part_of(g, q.grank, k, &off, &len);
        RET(ipc_fence2(g));
        PullBatch pb;
        RET(pull_begin(pb, g, q, q.cstream));
        for (int p = 0; p < g.P; ++p)
            if (p != q.grank && len)
                RET(ipc_pull(pb, p, at(q.R, (size_t)p * c + off, e), g.peerT[(size_t)p] + ((size_t)q.grank * c + off) * e, len * e));
        return pull_end(pb);
    }
    NCCLCHK(rc->GroupStart());
    for (ShardRank &q : g.r) {
        size_t offq, lenq;
        part_of(g, q.grank, k, &offq, &lenq);
        for (int peer = 0; peer < g.P; ++peer) {
            if (peer == q.grank) continue;
            size_t offp, lenp;
            part_of(g, peer, k, &offp, &lenp);
            if (lenp) NCCLCHK(rc->Send(at(q.T, (size_t)peer * c + offp, e), lenp, vtype, peer, q.comm2, q.cstream));
            if (lenq) NCCLCHK(rc->Recv(at(q.R, (size_t)peer * c + offq, e), lenq, vtype, peer, q.comm2, q.cstream));
        }
    }
    NCCLCHK(rc->GroupEnd());
    return LSQRHIP_OK;
}

// part k of every slice of v to everybody, on the exchange streams (after evV); records evAG[k]
static int ov_gather_part(ShardGroup &g, int k)
{
    Rccl *rc = rccl();
    const size_t c = (size_t)g.chunk, e = g.esz;
    const ncclDataType_t vtype = e == sizeof(float) ? ncclFloat : ncclDouble;
    if (g.loopback) {
        for (ShardRank &q : g.r) {
            PullBatch pb;
            RET(pull_begin(pb, g, q, q.cstream));
            for (ShardRank &p : g.r) {
                if (&p == &q) continue;
                size_t off, len;
                part_of(g, p.grank, k, &off, &len);
                if (len == 0) continue;
                RET(pull(pb, p, at(q.V, (size_t)p.grank * c + off, e), at(p.V, (size_t)p.grank * c + off, e), len * e));
            }
            RET(pull_end(pb));
            HIPCHK(hipEventRecord(q.evAG[(size_t)k], q.cstream));
        }
        return LSQRHIP_OK;
    }
    if (g.ipc) {   // (the exchange stream has waited for this rank's v_q: evV) every rank's v_q final -- one fence, in front
        ShardRank &q = g.r[0];   // of part 0 -- then part k of every peer's slice pulled
        if (k == 0) RET(ipc_fence2(g));
        PullBatch pb;
        RET(pull_begin(pb, g, q, q.cstream));
        for (int p = 0; p < g.P; ++p) {
            if (p == q.grank) continue;
            size_t off, len;
            part_of(g, p, k, &off, &len);
            if (len) RET(ipc_pull(pb, p, at(q.V, (size_t)p * c + off, e), g.peerV[(size_t)p] + ((size_t)p * c + off) * e, len * e));
        }
        RET(pull_end(pb));
        HIPCHK(hipEventRecord(q.evAG[(size_t)k], q.cstream));
        return LSQRHIP_OK;
    }
    NCCLCHK(rc->GroupStart());
    for (ShardRank &q : g.r) {
        size_t offq, lenq;
        part_of(g, q.grank, k, &offq, &lenq);
        for (int peer = 0; peer < g.P; ++peer) {
            if (peer == q.grank) continue;
            size_t offp, lenp;
            part_of(g, peer, k, &offp, &lenp);
            if (lenq) NCCLCHK(rc->Send(at(q.V, (size_t)q.grank * c + offq, e), lenq, vtype, peer, q.comm2, q.cstream));
            if (lenp) NCCLCHK(rc->Recv(at(q.V, (size_t)peer * c + offp, e), lenp, vtype, peer, q.comm2, q.cstream));
        }
    }
    NCCLCHK(rc->GroupEnd());
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        HIPCHK(hipEventRecord(q.evAG[(size_t)k], q.cstream));
    }
    return LSQRHIP_OK;
}

static int enqueue_iteration_overlap(ShardGroup &g)
{
    const int G = g.parts;
    // which products have a plan of G phases (column-swept layouts built under the switch): the others run whole
    bool phA = true, phT = true;
    for (ShardRank &q : g.r) {
        phA = phA && q.h->A.csb && q.h->A.phases == G;
        phT = phT && q.h->AT.csb && q.h->AT.phases == G;
        // Without the piece maxima of v in the norms' message (LSQRHIP_SHARD_VMAX=0, or a world of 69..128 ranks whose
        // message would not fit) phase 0 of mode 1 would take them with a pass over the WHOLE of V -- while the parts
        // 1..G-1 of it are still arriving: maxima of a mix of old and new v, grids that differ from run to run
        // (round-4 advisor).  Mode 1 then runs whole, behind all the parts: bit for bit the plain schedule again.
        phA = phA && q.h->shard.vmax_msg;
    }
    auto wait_all = [&](ShardRank &q, auto getev) -> int {   // q's compute stream waits for an exchange-stream event
        HIPCHK(hipSetDevice(q.h->device));
        for (ShardRank &p : g.r)
            if (&p == &q || g.loopback) HIPCHK(hipStreamWaitEvent(q.h->stream, getev(p), 0));
        return LSQRHIP_OK;
    };
    // mode 1, phase by phase behind the parts of v
    for (int k = 0; k < G; ++k) {
        for (ShardRank &q : g.r) RET(wait_all(q, [&](ShardRank &p) { return p.evAG[(size_t)k]; }));
        if (phA) RET(stage_all(g, ST_MODE1, k));
    }
    if (!phA) RET(stage_all(g, ST_MODE1));
    RET(ex_scalars(g, 1, false, true));
    // mode 2, its parts leaving as they complete
    for (int k = 0; k < G; ++k) {
        if (phT) RET(stage_all(g, ST_S1_ATU, k));
        else if (k == G - 1) RET(stage_all(g, ST_S1_ATU));
        if (phT || k == G - 1)
            for (int kk = phT ? k : 0; kk <= k; ++kk) {
                for (ShardRank &q : g.r) {
                    HIPCHK(hipSetDevice(q.h->device));
                    HIPCHK(hipEventRecord(q.evT[(size_t)kk], q.h->stream));
                }
                RET(ov_scatter_part(g, kk));
            }
    }
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        HIPCHK(hipEventRecord(q.evRS, q.cstream));
    }
    for (ShardRank &q : g.r) RET(wait_all(q, [&](ShardRank &p) { return p.evRS; }));
    RET(stage_all(g, ST_VCOMBINE));
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        HIPCHK(hipEventRecord(q.evV, q.h->stream));
    }
    // the gather of v, part by part on the exchange streams, while the compute streams finish the iteration
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        for (ShardRank &p : g.r)
            if (&p == &q || g.loopback) HIPCHK(hipStreamWaitEvent(q.cstream, p.evV, 0));
    }
    for (int k = 0; k < G; ++k) RET(ov_gather_part(g, k));
    RET(ex_scalars(g, 2, false, true));
    RET(stage_all(g, ST_UPDATE));
    return LSQRHIP_OK;
}

// `count` iterations: the stages of every local rank and the three exchanges of each iteration, all asynchronous
static int enqueue_iterations(ShardGroup &g, int count)
{
    for (int k = 0; k < count; ++k) {
        if (g.overlap) {
            RET(enqueue_iteration_overlap(g));
            continue;
        }
        RET(stage_all(g, ST_MODE1));
        RET(ex_scalars(g, 1, false, true));
        RET(stage_all(g, ST_S1_ATU));
        RET(ex_scatter(g));
        RET(stage_all(g, ST_VCOMBINE));
        RET(ex_scalars(g, 2, true, true));    // alpha^2, dknorm^2 and the v slices in one group
        RET(stage_all(g, ST_UPDATE));
    }
    return LSQRHIP_OK;
}

// One hipGraph per batch of `poll_every` iterations, as solve_loop.h does for one GPU: ~14 launches, 2 copies and
// 3 exchanges per iteration become one graph launch per batch (iterations past the stop are no-ops: every kernel
// looks at the flag first).  Only for a group with ONE local rank -- a world of 1, or one rank of a
// one-process-per-GPU world: the capture is a plain single-stream capture.  (Forking the other ranks' streams of
// the loopback harness into the capture and joining them back crashed inside the HIP runtime on ROCm 7.2 -- several
// ranks in one process stay eager, as does one process driving several devices over RCCL.)
//   LSQRHIP_SHARD_GRAPH   0 never | 1 also with RCCL between processes (RCCL's kernels are captured like any
//                         others) | unset: only where no RCCL call is involved (world of 1) -- the RCCL form stays
//                         eager until a multi-GPU node has run it captured (tests/test_gpu_engine.py::test_rccl_*)
static bool group_graph_wanted(const ShardGroup &g)
{
    const int mode = env_int("LSQRHIP_SHARD_GRAPH", -1);   // (read at every solve: the tests switch it)
    if (mode == 0 || g.graph_state < 0 || g.r.size() != 1 || g.overlap || g.ipc) return false;   // (ipc: copy streams fork)
    return g.P == 1 || mode == 1;
}

static int capture_group_batch(ShardGroup &g)
{
    ShardRank &q0 = g.r[0];
    hipStream_t s0 = q0.h->stream;
    HIPCHK(hipSetDevice(q0.h->device));
    if (g.gexec) {
        (void)hipGraphExecDestroy(g.gexec);
        g.gexec = nullptr;
    }
    HIPCHK(hipStreamSynchronize(s0));
    hipError_t e = hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return fail(LSQRHIP_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    const int rc = enqueue_iterations(g, g.poll_every);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(s0, &graph);
    if (e != hipSuccess || rc != LSQRHIP_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return fail(LSQRHIP_ERR_HIP, std::string("capture of a sharded batch failed: ") +
                                         (e != hipSuccess ? hipGetErrorString(e) : g_last_error.c_str()));
    }
    e = hipGraphInstantiate(&g.gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        g.gexec = nullptr;
        return fail(LSQRHIP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    g.gexec_epoch.assign(1, q0.h->graph_epoch);
    return LSQRHIP_OK;
}

// The loop.  b: every local rank's block is in q.bloc.  Outputs: x (and se) replicated in q.xfull / q.sefull.
static int run_group_body(ShardGroup &g, double damp, double atol, double btol, double conlim, int itnlim, int wantse,
                     int *istop, int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (g.P > 1 && !g.loopback && !rccl()) return fail(LSQRHIP_ERR_HIP, "librccl.so.1 could not be loaded");
    if (g.overlap) {
        RET(overlap_setup(g));
        for (ShardRank &q : g.r) {   // (what a previous solve left running on the exchange streams reads V, T: let it finish)
            HIPCHK(hipSetDevice(q.h->device));
            HIPCHK(hipStreamSynchronize(q.cstream));
        }
    }
    // v's piece maxima ride with the norms when mode 1 of this rank's block wants them (column-swept row blocks) --
    // a property of the local matrix only, but the message length must be the world's: every rank of a sharded
    // system built by one rule from one matrix takes the same layout family; LSQRHIP_SHARD_VMAX=0 / 1 forces it
    {
        const int force = env_int("LSQRHIP_SHARD_VMAX", -1);
        bool want = force < 0 ? (int64_t)g.P * SHARD_NMAX <= (int64_t)CSB_XMAX_GRID * (VEC_BLOCK / WAVE) : force != 0;
        if ((int64_t)g.P * SHARD_NMAX > (int64_t)CSB_XMAX_GRID * (VEC_BLOCK / WAVE)) want = false;
        g.msg = want ? SHARD_MSG : 4;
        for (ShardRank &q : g.r) {
            q.h->shard.vmax_msg = want;
            q.h->shard.own_in_T = true;
            q.h->shard.gath = q.gath;
            q.h->shard.msg = g.msg;
            q.h->shard.engine_next = true;   // (lsqrhip_shard_begin: these fields are meant)
        }
    }
    for (ShardRank &q : g.r)
        RET(lsqrhip_shard_begin(q.h, q.bloc, g.m, g.P, q.grank, damp, atol, btol, conlim, itnlim, wantse, q.T, q.R, q.V,
                                q.sums));
    RET(stage_all(g, ST_SUMSQ_B));
    RET(ex_scalars(g, 3));
    RET(stage_all(g, ST_INIT_BETA_ATU));
    RET(ex_scatter(g));
    RET(stage_all(g, ST_INIT_V));
    RET(ex_scalars(g, 2));
    RET(stage_all(g, ST_INIT_W));
    RET(ex_gather(g, false, false));
    if (g.overlap) {   // the first iteration's mode 1 waits for "its" parts of v: they are all there
        for (ShardRank &q : g.r) {
            HIPCHK(hipSetDevice(q.h->device));
            for (int k = 0; k < g.parts; ++k) HIPCHK(hipEventRecord(q.evAG[(size_t)k], q.h->stream));
        }
    }
    int st3[3] = {0, 0, 0};
    auto poll = [&]() -> int {   // the scalar recurrences are replicated bit for bit: every rank sees the same flag
        for (ShardRank &q : g.r) RET(lsqrhip_shard_poll(q.h, st3));
        return LSQRHIP_OK;
    };
    RET(poll());
    // batches: one graph launch each where the exchanges can be captured (group_graph_wanted), else eager
    bool graph = group_graph_wanted(g);
    if (graph) {
        bool stale = g.gexec == nullptr || g.gexec_epoch.size() != g.r.size();
        for (size_t i = 0; !stale && i < g.r.size(); ++i) stale = g.gexec_epoch[i] != g.r[i].h->graph_epoch;
        if (stale && capture_group_batch(g) != LSQRHIP_OK) {   // never fatal: the eager form is always there
            g.graph_state = -1;
            graph = false;
        } else {
            g.graph_state = 1;
        }
    }
    int64_t launched = 0;
    const int fail_at = env_int("LSQRHIP_SHARD_FAIL_AT", -1);   // test hook: an error exit after this many queued iterations
    while (!st3[0]) {
        if (fail_at >= 0 && launched >= fail_at)
            return fail(LSQRHIP_ERR_HIP, "LSQRHIP_SHARD_FAIL_AT: injected failure of the sharded engine (test hook)");
        if (launched > (int64_t)itnlim + g.poll_every)
            return fail(LSQRHIP_ERR_HIP, "sharded iteration loop did not terminate (device state not advancing)");
        if (graph) {
            HIPCHK(hipSetDevice(g.r[0].h->device));
            HIPCHK(hipGraphLaunch(g.gexec, g.r[0].h->stream));
            launched += g.poll_every;
        } else {
            const int batch = (int)std::min<int64_t>(g.poll_every, std::max<int64_t>(1, (int64_t)itnlim - launched));
            RET(enqueue_iterations(g, batch));
            launched += batch;
        }
        RET(poll());
    }
    if (g.overlap)   // (iterations enqueued past the stop still exchanged: drain them before x is assembled)
        for (ShardRank &q : g.r) {
            HIPCHK(hipSetDevice(q.h->device));
            HIPCHK(hipStreamSynchronize(q.cstream));
        }
    for (ShardRank &q : g.r) {
        int is = 0, it = 0;
        double sc[5];
        RET(lsqrhip_shard_end(q.h, q.xfull, wantse ? q.sefull : nullptr, &is, &it, sc, sc + 1, sc + 2, sc + 3, sc + 4));
        if (&q == &g.r[0]) {
            if (istop) *istop = is;
            if (itn) *itn = it;
            if (anorm) *anorm = sc[0];
            if (acond) *acond = sc[1];
            if (rnorm) *rnorm = sc[2];
            if (arnorm) *arnorm = sc[3];
            if (xnorm) *xnorm = sc[4];
        }
    }
    RET(ex_gather(g, true, wantse != 0));
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        HIPCHK(hipStreamSynchronize(q.h->stream));
    }
    return LSQRHIP_OK;
}

// ... and whatever way it ends, the handles are left as a caller of the C stages expects them: an error exit above (an RCCL
// call that failed, a poll error, the "did not terminate" guard, LSQRHIP_SHARD_FAIL_AT in the tests) never reached
// lsqrhip_shard_end, and the engine-only fields of the ranks' handles would still say "the norms are gathered in gath, your
// own slice stays in T, sums is a 64-double message" to the Python stage driver that lsqr_amd/dist_bench.py falls back to.
static int run_group(ShardGroup &g, double damp, double atol, double btol, double conlim, int itnlim, int wantse,
                     int *istop, int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    const int rc = run_group_body(g, damp, atol, btol, conlim, itnlim, wantse, istop, itn, anorm, acond, rnorm, arnorm, xnorm);
    if (rc != LSQRHIP_OK) {
        const std::string why = g_last_error;
        for (ShardRank &q : g.r) {
            if (q.h == nullptr) continue;
            (void)hipSetDevice(q.h->device);
            (void)hipStreamSynchronize(q.h->stream);   // (stages already queued read gath / T: let them finish first)
            if (q.cstream) (void)hipStreamSynchronize(q.cstream);
            ShardCtx &c = q.h->shard;
            c.active = false;
            c.own_in_T = false;
            c.gath = nullptr;
            c.vmax_msg = false;
            c.msg = 4;
            c.engine_next = false;
        }
        (void)hipGetLastError();
        g_last_error = why;
    }
    return rc;
}

// Every rank of the world uses the largest norm_exp (scalar.h "range-safe norms").
static int agree_norm_exp(ShardGroup &g)
{
    int e = -100000;
    for (ShardRank &q : g.r) e = std::max(e, q.h->norm_exp);
    if (g.P > (int)g.r.size()) {   // other processes: one tiny all-gather of ints (as doubles)
        Rccl *rc = rccl();
        for (ShardRank &q : g.r) {
            HIPCHK(hipSetDevice(q.h->device));
            double v[4] = {(double)e, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(q.sums, v, sizeof(v), hipMemcpyHostToDevice, q.h->stream));
        }
        NCCLCHK(rc->GroupStart());
        for (ShardRank &q : g.r) NCCLCHK(rc->AllGather(q.sums, q.gath, 4, ncclDouble, q.comm, q.h->stream));
        NCCLCHK(rc->GroupEnd());
        std::vector<double> all(4 * (size_t)g.P);
        ShardRank &q0 = g.r[0];
        HIPCHK(hipSetDevice(q0.h->device));
        HIPCHK(hipMemcpyAsync(all.data(), q0.gath, sizeof(double) * all.size(), hipMemcpyDeviceToHost, q0.h->stream));
        HIPCHK(hipStreamSynchronize(q0.h->stream));
        for (int r = 0; r < g.P; ++r) e = std::max(e, (int)all[4 * (size_t)r]);
    }
    for (ShardRank &q : g.r) RET(lsqrhip_set_option(q.h, "norm_exp", e));
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// one process, ngpu devices
// ---------------------------------------------------------------------------------------------------
// AT = double: binary64 sub-handles (lsqrhip_create); AT = float: REAL32 sub-handles (lsqrhip_create_f32: real32
// storage on the devices and real32 slices on the links, or binary64 with LSQRHIP_REAL32_MIXED=1)
template <typename AT>
static int create_sharded_T(int m, int n, int64_t nnz, const int *irow, const int *icol, const AT *a, int ngpu,
                            lsqrhip_handle_t *out)
{
    if (!out) return fail(LSQRHIP_ERR_ARG, "null handle pointer");
    *out = nullptr;
    if (ngpu < 1) return fail(LSQRHIP_ERR_ARG, "ngpu must be >= 1");
    if (m < 0 || n < 0 || nnz < 0) return fail(LSQRHIP_ERR_ARG, "negative dimension");
    const int ngpu_asked = ngpu;
    if (nnz > 0 && (!irow || !icol || !a)) return fail(LSQRHIP_ERR_SIZES, lsqrhip_error_string(LSQRHIP_ERR_SIZES));
    const int have = lsqrhip_device_count();
    // test harness: more ranks than devices, exchanges as device copies inside the process (see fence_ranks)
    // LSQRHIP_SHARD_COPY=1: the same copies between the DEVICES of the node (one rank per device, peer copies -- the
    // SDMA engines -- instead of RCCL's send / receive kernels: no CU set aside for an exchange; opt-in, not timed yet)
    const bool copy_mode = env_int("LSQRHIP_SHARD_COPY", 0) != 0 && have >= std::min(ngpu, std::max(m, 1));
    const bool harness = env_int("LSQRHIP_SHARD_LOOPBACK", 0) != 0 && have >= 1;
    const bool loopback = harness || copy_mode;
    if (have < ngpu && !loopback)
        return fail(LSQRHIP_ERR_NO_DEVICE, "ngpu = " + std::to_string(ngpu) + " but this node shows " + std::to_string(have) +
                                               " usable gfx950 device(s)");
    ngpu = std::min(ngpu_asked, std::max(m, 1));  // never more row blocks than rows
    if (!harness && g_device.load() + ngpu > have)   // blocks go to devices [selected, selected + ngpu)
        return fail(LSQRHIP_ERR_NO_DEVICE, "ngpu = " + std::to_string(ngpu) + " starting at the selected device " +
                                               std::to_string(g_device.load()) + " exceeds the " + std::to_string(have) +
                                               " device(s) of this node (lsqrhip_set_device)");
    if (ngpu > 1 && !loopback && !rccl()) return fail(LSQRHIP_ERR_HIP, "librccl.so.1 could not be loaded");
    // One pass over the triplets: the reference's checks (src/lsqr.f90:110-111) on the whole system, and the row
    // counts.  Then contiguous row blocks balanced by nonzeros (+1 per row: a row costs work even when empty).
    std::vector<int64_t> cum((size_t)m + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) {
        if (irow[k] < 1 || irow[k] > m) return fail(LSQRHIP_ERR_IROW, lsqrhip_error_string(LSQRHIP_ERR_IROW));
        if (icol[k] < 1 || icol[k] > n) return fail(LSQRHIP_ERR_ICOL, lsqrhip_error_string(LSQRHIP_ERR_ICOL));
        cum[(size_t)irow[k]] += 1;
    }
    for (int r = 0; r < m; ++r) cum[(size_t)r + 1] += cum[(size_t)r] + 1;
    std::vector<int> cut((size_t)ngpu + 1, 0);
    cut[(size_t)ngpu] = m;
    for (int p = 1; p < ngpu; ++p) {
        const int64_t target = cum[(size_t)m] * p / ngpu;
        int r = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        r = std::min(std::max(r, cut[(size_t)p - 1] + (m >= ngpu ? 1 : 0)), m - (m >= ngpu ? ngpu - p : 0));
        cut[(size_t)p] = std::max(r, cut[(size_t)p - 1]);
    }
    // nonzeros of block p = what cum says of its rows, less the +1 per row
    std::vector<int64_t> first((size_t)ngpu + 1, 0);
    for (int p = 0; p < ngpu; ++p)
        first[(size_t)p + 1] = first[(size_t)p] + (cum[(size_t)cut[(size_t)p + 1]] - cum[(size_t)cut[(size_t)p]]) -
                               (cut[(size_t)p + 1] - cut[(size_t)p]);
    cum.clear();
    cum.shrink_to_fit();
    std::vector<unsigned char> owner8;
    std::vector<int> owner;
    // (a byte per row suffices for any node; the int form is for the loopback harness with > 255 ranks)
    if (ngpu <= 255) owner8.assign((size_t)std::max(m, 1), 0);
    else owner.assign((size_t)std::max(m, 1), 0);
    for (int p = 0; p < ngpu; ++p)
        for (int r = cut[(size_t)p]; r < cut[(size_t)p + 1]; ++r) {
            if (ngpu <= 255) owner8[(size_t)r] = (unsigned char)p;
            else owner[(size_t)r] = p;
        }
    // Triplets that already come block by block (row-sorted input, what every generator and most callers hand over):
    // block p is the range [first[p], first[p + 1]) of the caller's arrays as it stands -- only its row indices need a
    // copy (rebased to the block), one block at a time: 4 bytes per nonzero of the LARGEST block instead of 16 bytes
    // per nonzero of the whole system held through every block's create (16 GB at configs[3]).
    bool blockwise = true;
    {
        int last = 0;
        for (int64_t k = 0; k < nnz && blockwise; ++k) {
            const int r = irow[k] - 1;
            const int p = ngpu <= 255 ? (int)owner8[(size_t)r] : owner[(size_t)r];
            blockwise = p >= last;
            last = p;
        }
    }
    // ... otherwise ONE pass that files every triplet under its block, COO order kept inside each (the reference's
    // row sums are formed in that order, src/lsqr.f90:168-172): O(nnz) host work whatever ngpu is
    // (uninitialised storage: value-initialising 16 bytes per nonzero first would cost as much as the pass itself)
    std::unique_ptr<int[]> lr, lc;
    std::unique_ptr<AT[]> la;
    if (blockwise) {
        int64_t big = 1;
        for (int p = 0; p < ngpu; ++p) big = std::max(big, first[(size_t)p + 1] - first[(size_t)p]);
        lr.reset(new int[(size_t)big]);
    } else {
        const size_t cap = (size_t)std::max<int64_t>(nnz, 1);
        lr.reset(new int[cap]);
        lc.reset(new int[cap]);
        la.reset(new AT[cap]);
        std::vector<int64_t> pos(first.begin(), first.end() - 1);
        for (int64_t k = 0; k < nnz; ++k) {
            const int r = irow[k] - 1;
            const int p = ngpu <= 255 ? (int)owner8[(size_t)r] : owner[(size_t)r];
            const int64_t w = pos[(size_t)p]++;
            lr[(size_t)w] = irow[k] - cut[(size_t)p];
            lc[(size_t)w] = icol[k];
            la[(size_t)w] = a[k];
        }
    }
    owner8.clear(); owner8.shrink_to_fit();
    owner.clear(); owner.shrink_to_fit();

    H *h = nullptr;
    const int dev0 = g_device.load();
    RET(new_handle(m, n, 0, &h));  // the parent: dimensions, error state, the group
    h->nnz = nnz;
    ShardGroup *g = new ShardGroup();
    h->group = g;
    g->P = ngpu;
    g->m = m;
    g->n = n;
    g->chunk = ((int64_t)n + ngpu - 1) / ngpu;
    g->owned = true;
    g->loopback = loopback;
    g->copy_streams = env_int("LSQRHIP_SHARD_COPY_STREAMS", 1) != 0;
    g->r.resize((size_t)ngpu);
    int rc = LSQRHIP_OK;
    for (int p = 0; p < ngpu && rc == LSQRHIP_OK; ++p) {
        const int64_t f = first[(size_t)p], np = first[(size_t)p + 1] - f;
        ShardRank &q = g->r[(size_t)p];
        q.grank = p;
        q.row0 = cut[(size_t)p];
        // the block's device, for this thread's create only (never through the process-wide selection)
        t_device_override = harness ? dev0 + p % have : dev0 + p;
        t_shard_world = ngpu;   // (the overlap plan of the block's layouts, if asked for: lsqrhip.hip finish_create)
        if (blockwise) {
            for (int64_t k = 0; k < np; ++k) lr[(size_t)k] = irow[f + k] - cut[(size_t)p];
            rc = create_block(cut[(size_t)p + 1] - cut[(size_t)p], n, np, lr.get(), icol + f, a + f, &q.h);
        } else {
            rc = create_block(cut[(size_t)p + 1] - cut[(size_t)p], n, np, lr.get() + f, lc.get() + f, la.get() + f, &q.h);
        }
        t_device_override = -1;
        t_shard_world = 0;
    }
    lr.reset();
    lc.reset();
    la.reset();
    if (rc == LSQRHIP_OK)
        for (ShardRank &q : g->r)
            if ((rc = alloc_rank_buffers(*g, q)) != LSQRHIP_OK) break;
    if (rc == LSQRHIP_OK && ngpu > 1 && !loopback) {
        std::vector<int> devs;
        std::vector<ncclComm_t> comms((size_t)ngpu);
        for (ShardRank &q : g->r) devs.push_back(q.h->device);
        ncclResult_t nr = rccl()->CommInitAll(comms.data(), ngpu, devs.data());
        if (nr != ncclSuccess) rc = fail(LSQRHIP_ERR_HIP, std::string("ncclCommInitAll: ") + rccl()->GetErrorString(nr));
        else
            for (int p = 0; p < ngpu; ++p) g->r[(size_t)p].comm = comms[(size_t)p];
    }
    // LSQRHIP_SHARD_OVERLAP=1: the n-vector exchanges in parts beside the products (enqueue_iteration_overlap); the
    // blocks' layouts were built for it above (t_shard_world).  Over RCCL it needs a second communicator.
    if (rc == LSQRHIP_OK && ngpu > 1 && env_int("LSQRHIP_SHARD_OVERLAP", 0) != 0) {
        g->overlap = 1;
        g->parts = std::min(std::max(env_int("LSQRHIP_SHARD_PARTS", 2), 2), 4);
        if (!loopback) {
            std::vector<int> devs;
            std::vector<ncclComm_t> comms((size_t)ngpu);
            for (ShardRank &q : g->r) devs.push_back(q.h->device);
            ncclResult_t nr = rccl()->CommInitAll(comms.data(), ngpu, devs.data());
            if (nr != ncclSuccess) g->overlap = 0;   // (never fatal: the plain schedule is always there)
            else
                for (int p = 0; p < ngpu; ++p) g->r[(size_t)p].comm2 = comms[(size_t)p];
        }
    }
    if (rc == LSQRHIP_OK) rc = agree_norm_exp(*g);
    if (rc != LSQRHIP_OK) {
        std::string keep = g_last_error;
        lsqrhip_destroy(h);
        g_last_error = keep;
        return rc;
    }
    (void)hipSetDevice(dev0);
    h->io32 = sizeof(AT) == sizeof(float);
    h->f32 = h->io32 && !g->r.empty() && g->r[0].h->f32;   // (false in the mixed mode: binary64 on the devices)
    *out = h;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_create_sharded(int m, int n, int64_t nnz, const int *irow, const int *icol, const double *a,
                                      int ngpu, lsqrhip_handle_t *out)
{
    return create_sharded_T<double>(m, n, nnz, irow, icol, a, ngpu, out);
}

extern "C" int lsqrhip_create_sharded_f32(int m, int n, int64_t nnz, const int *irow, const int *icol, const float *a,
                                          int ngpu, lsqrhip_handle_t *out)
{
    return create_sharded_T<float>(m, n, nnz, irow, icol, a, ngpu, out);
}

// lsqrhip_solve on a sharded handle: b is cut into the row blocks, x (se) come back whole.
static int solve_group_host(H *h, const double *b, double damp, double atol, double btol, double conlim, int itnlim,
                            int wantse, int want_log, double *x, double *se, int *istop, int *itn, double *anorm,
                            double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    ShardGroup &g = *h->group;
    if (!istop || (!x && g.n > 0) || (!b && g.m > 0)) return fail(LSQRHIP_ERR_ARG, "null b, x or istop");
    if (wantse && !se) return fail(LSQRHIP_ERR_ARG, "wantse set but se is null");
    // the iteration log: rank 0 keeps the records (replicated scalars; x(1) is the first entry of its slice)
    for (ShardRank &q : g.r) q.h->shard.want_log = (want_log != 0 && &q == &g.r[0]) ? 1 : 0;
    // (b, x, se: float arrays in disguise for a REAL32 group -- lsqrhip_solve_f32 -- like the vectors on the devices)
    for (ShardRank &q : g.r) {
        HIPCHK(hipSetDevice(q.h->device));
        if (q.h->m > 0)
            HIPCHK(hipMemcpyAsync(q.bloc, reinterpret_cast<const char *>(b) + g.esz * (size_t)q.row0, g.esz * (size_t)q.h->m,
                                  hipMemcpyHostToDevice, q.h->stream));
    }
    RET(run_group(g, damp, atol, btol, conlim, itnlim, wantse, istop, itn, anorm, acond, rnorm, arnorm, xnorm));
    ShardRank &q0 = g.r[0];
    HIPCHK(hipSetDevice(q0.h->device));
    if (g.n > 0) HIPCHK(hipMemcpy(x, q0.xfull, g.esz * (size_t)g.n, hipMemcpyDeviceToHost));
    if (wantse && g.n > 0) HIPCHK(hipMemcpy(se, q0.sefull, g.esz * (size_t)g.n, hipMemcpyDeviceToHost));
    return LSQRHIP_OK;
}

// lsqrhip_aprod on a sharded handle (host vectors): the blocks one after the other
static int aprod_block(H *h, int mode, double *x, double *y) { return lsqrhip_aprod(h, mode, x, y); }
static int aprod_block(H *h, int mode, float *x, float *y) { return lsqrhip_aprod_f32(h, mode, x, y); }

template <typename VT>
static int aprod_group_host_T(H *h, int mode, VT *x, VT *y)
{
    ShardGroup &g = *h->group;
    if (mode == 1) {
        for (ShardRank &q : g.r) RET(aprod_block(q.h, 1, x, y + q.row0));
        return LSQRHIP_OK;
    }
    std::vector<VT> t((size_t)std::max(g.n, 1));
    std::vector<double> acc((size_t)std::max(g.n, 1), 0.0);
    for (ShardRank &q : g.r) {   // x += sum_p A_p' y_p, partial products added in rank order
        std::fill(t.begin(), t.end(), (VT)0);
        RET(aprod_block(q.h, 2, t.data(), y + q.row0));
        for (int j = 0; j < g.n; ++j)
            acc[(size_t)j] = &q == &g.r[0] ? (double)t[(size_t)j] : acc[(size_t)j] + (double)t[(size_t)j];
    }
    for (int j = 0; j < g.n; ++j) x[j] = (VT)((double)x[j] + acc[(size_t)j]);
    return LSQRHIP_OK;
}
static int aprod_group_host(H *h, int mode, double *x, double *y) { return aprod_group_host_T<double>(h, mode, x, y); }
static int aprod_group_host_f32(H *h, int mode, float *x, float *y) { return aprod_group_host_T<float>(h, mode, x, y); }

// ---------------------------------------------------------------------------------------------------
// one process per GPU
// ---------------------------------------------------------------------------------------------------
extern "C" int lsqrhip_rccl_unique_id(char *out128)
{
    if (!out128) return fail(LSQRHIP_ERR_ARG, "null buffer");
    Rccl *rc = rccl();
    if (!rc) return fail(LSQRHIP_ERR_HIP, "librccl.so.1 could not be loaded");
    ncclUniqueId id;
    NCCLCHK(rc->GetUniqueId(&id));
    std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_shard_comm_init(lsqrhip_handle_t h, int world, int rank, int64_t row0, int64_t m_global,
                                       const char *id128)
{
    if (!h || h->op || h->group || h->mp)
        return fail(LSQRHIP_ERR_ARG, "needs a matrix handle that is not yet part of a group or a world");
    if (world < 1 || rank < 0 || rank >= world) return fail(LSQRHIP_ERR_ARG, "bad world / rank");
    if (world > 1 && !id128) return fail(LSQRHIP_ERR_ARG, "null unique id");
    if (world > 1 && !rccl()) return fail(LSQRHIP_ERR_HIP, "librccl.so.1 could not be loaded");
    HIPCHK(hipSetDevice(h->device));
    ShardGroup *g = new ShardGroup();
    g->P = world;
    g->m = (int)m_global;
    g->n = h->n;
    g->chunk = ((int64_t)h->n + world - 1) / world;
    g->owned = false;
    g->r.resize(1);
    ShardRank &q = g->r[0];
    q.h = h;
    q.grank = rank;
    q.row0 = row0;
    int rc = alloc_rank_buffers(*g, q);
    if (rc == LSQRHIP_OK && world > 1) {
        ncclUniqueId id;
        std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t nr = rccl()->CommInitRank(&q.comm, world, id, rank);
        if (nr != ncclSuccess) rc = fail(LSQRHIP_ERR_HIP, std::string("ncclCommInitRank: ") + rccl()->GetErrorString(nr));
    }
    // LSQRHIP_SHARD_OVERLAP=1 (set on EVERY rank, before the handle was created: its layouts are built for it with
    // LSQRHIP_SHARD_WORLD = world): a second communicator for the exchange stream, split off the first
    if (rc == LSQRHIP_OK && world > 1 && env_int("LSQRHIP_SHARD_OVERLAP", 0) != 0) {
        if (rccl()->CommSplit == nullptr) {
            rc = fail(LSQRHIP_ERR_HIP, "LSQRHIP_SHARD_OVERLAP=1 needs ncclCommSplit, which this librccl does not export");
        } else {
            ncclResult_t nr = rccl()->CommSplit(q.comm, 0, rank, &q.comm2, nullptr);
            if (nr != ncclSuccess) rc = fail(LSQRHIP_ERR_HIP, std::string("ncclCommSplit: ") + rccl()->GetErrorString(nr));
            g->overlap = 1;
            g->parts = std::min(std::max(env_int("LSQRHIP_SHARD_PARTS", 2), 2), 4);
        }
    }
    // LSQRHIP_SHARD_COPY=1 (set on EVERY rank): the n-vector exchanges as copies over IPC-mapped buffers (ipc_setup) -- of
    // the plain schedule and of the overlapped one (whose parts are then fenced by 8-byte all-gathers on the second
    // communicator: the only RCCL kernels left beside the sweeps are one-workgroup ones)
    if (rc == LSQRHIP_OK && world > 1 && env_int("LSQRHIP_SHARD_COPY", 0) != 0) rc = ipc_setup(*g);
    if (rc == LSQRHIP_OK) rc = agree_norm_exp(*g);
    if (rc != LSQRHIP_OK) {
        q.h = nullptr;  // not ours to destroy
        free_group(g);
        return rc;
    }
    h->mp = g;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_shard_solve(lsqrhip_handle_t h, const double *d_b_local, double damp, double atol, double btol,
                                   double conlim, int itnlim, int wantse, double *d_x, double *d_se, int *istop, int *itn,
                                   double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (!h || !h->mp) return fail(LSQRHIP_ERR_NOT_INIT, "lsqrhip_shard_comm_init was not called");
    ShardGroup &g = *h->mp;
    ShardRank &q = g.r[0];
    HIPCHK(hipSetDevice(h->device));
    if (h->m > 0) {
        if (!d_b_local) return fail(LSQRHIP_ERR_ARG, "null b");
        HIPCHK(hipMemcpyAsync(q.bloc, d_b_local, g.esz * (size_t)h->m, hipMemcpyDeviceToDevice, h->stream));
    }
    RET(run_group(g, damp, atol, btol, conlim, itnlim, wantse, istop, itn, anorm, acond, rnorm, arnorm, xnorm));
    if (d_x && g.n > 0) HIPCHK(hipMemcpy(d_x, q.xfull, g.esz * (size_t)g.n, hipMemcpyDeviceToDevice));
    if (d_se && wantse && g.n > 0) HIPCHK(hipMemcpy(d_se, q.sefull, g.esz * (size_t)g.n, hipMemcpyDeviceToDevice));
    return LSQRHIP_OK;
}
